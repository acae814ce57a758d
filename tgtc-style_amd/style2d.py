"""2-D style pass on the HIP library (include/tgtc_style2d.h): patch embedding, style transformer, CNN
decoder, VGG encoder, calc_mean_std / AdaIN and the trans_test.py post-processing.

The modules keep the reference's class names, constructor defaults and state-dict key names
(tctrans.py:13-33 PatchEmbed, :36-66 decoder, :68-99 vgg, :138-245 StyTrans; transformer.py:13-75
Transformer; function.py:4-12 calc_mean_std; Style_function.py:15-24 adaptive_instance_normalization) so
reference checkpoints load with `load_state_dict`; forward runs the HIP kernels (eval mode, batch of 1).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import hip

c_void_p, c_int, c_int64, c_float, c_size_t, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                         ctypes.c_float, ctypes.c_size_t, ctypes.c_char_p)


class NamedTensor(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", c_void_p), ("numel", c_int64)]


_NT = ctypes.POINTER(NamedTensor)
SIGNATURES = {
    "tgtc_s2d_create": [_NT, c_int, _NT, c_int, _NT, c_int, _NT, c_int, c_int, ctypes.POINTER(c_void_p)],
    "tgtc_s2d_destroy": [c_void_p],
    "tgtc_s2d_patch_embed": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "tgtc_s2d_transformer_workspace_bytes": [c_int, c_int],
    "tgtc_s2d_transformer_forward": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    "tgtc_s2d_mha": [c_void_p, c_char_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    "tgtc_s2d_encoder_layer": [c_void_p, c_char_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    "tgtc_s2d_decoder_layer": [c_void_p, c_char_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                               c_void_p, c_void_p],
    "tgtc_s2d_decode_workspace_bytes": [c_int, c_int],
    "tgtc_s2d_cnn_decode": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    "tgtc_s2d_vgg_workspace_bytes": [c_int, c_int],
    "tgtc_s2d_vgg_encode": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p],
    "tgtc_s2d_mean_std": [c_void_p, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p],
    "tgtc_s2d_linear": [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "tgtc_s2d_split": [c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    "tgtc_s2d_linear_pre": [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "tgtc_s2d_linear_backward_workspace_bytes": [c_int64, c_int, c_int],
    "tgtc_s2d_linear_backward": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                 c_void_p, c_void_p, c_void_p],
    "tgtc_s2d_activation": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    "tgtc_s2d_adain": [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    "tgtc_s2d_resize_bilinear": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "tgtc_s2d_style_feature": [c_void_p, c_int, c_void_p, c_void_p],
    "tgtc_s2d_tokens_to_nchw": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "tgtc_s2d_nchw_to_tokens": [c_void_p, c_int, c_int, c_void_p, c_void_p],
}
RESTYPES = {k: c_size_t for k in SIGNATURES if k.endswith("_workspace_bytes")}
hip.register(SIGNATURES, RESTYPES)


def _named(state):
    if not state:
        return None, 0, []
    arr = (NamedTensor * len(state))()
    keep = []
    for i, (k, v) in enumerate(state.items()):
        a = np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
        name = k.encode()
        keep += [a, name]
        arr[i].name, arr[i].data, arr[i].numel = name, a.ctypes.data, a.size
    return arr, len(state), keep


class Handle:
    """Owns a tgtc_style2d handle (device copies of the parameter groups given)."""

    def __init__(self, transformer=None, embedding=None, decoder=None, vgg=None, precision="fp16x3"):
        hip.require_gpu()
        lib = hip.load()
        t, nt, k1 = _named(transformer)
        e, ne, k2 = _named(embedding)
        d, nd, k3 = _named(decoder)
        v, nv, k4 = _named(vgg)
        h = c_void_p()
        hip.check(lib.tgtc_s2d_create(t, nt, e, ne, d, nd, v, nv, hip.PRECISIONS[precision], ctypes.byref(h)))
        self.handle, self.precision = h, precision
        self._ws = None

    def __del__(self):
        try:
            if self.handle:
                hip.load().tgtc_s2d_destroy(self.handle)
        except Exception:
            pass
        self.handle = None

    def workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
        return self._ws

    # ---- ops (inputs: contiguous float32 CUDA tensors)
    def patch_embed(self, img):
        """img [1,3,H,W] -> tokens [(H//8)*(W//8), 512]"""
        lib = hip.load()
        _, _, H, W = img.shape
        img = img.float().contiguous()
        tok = torch.empty((H // 8) * (W // 8), 512, device=img.device)
        hip.check(lib.tgtc_s2d_patch_embed(self.handle, hip.ptr(img), H, W, hip.ptr(tok), hip.stream()))
        return tok

    def transformer(self, style_tokens, content_tokens):
        lib = hip.load()
        ns, nc = style_tokens.shape[0], content_tokens.shape[0]
        ws = self.workspace(lib.tgtc_s2d_transformer_workspace_bytes(ns, nc))
        s, c = style_tokens.float().contiguous(), content_tokens.float().contiguous()
        hs = torch.empty(nc, 512, device=s.device)
        hip.check(lib.tgtc_s2d_transformer_forward(self.handle, hip.ptr(s), ns, hip.ptr(c), nc, hip.ptr(ws), ws.numel(),
                                                   hip.ptr(hs), hip.stream()))
        return hs

    def _layer_ws(self, L, S):
        return self.workspace(hip.load().tgtc_s2d_transformer_workspace_bytes(max(L, S), max(L, S)))

    def mha(self, prefix, q, k, v):
        lib = hip.load()
        q, k, v = (t.float().contiguous() for t in (q, k, v))
        ws = self._layer_ws(q.shape[0], k.shape[0])
        out = torch.empty(q.shape[0], 512, device=q.device)
        hip.check(lib.tgtc_s2d_mha(self.handle, prefix.encode(), hip.ptr(q), q.shape[0], hip.ptr(k), hip.ptr(v),
                                   k.shape[0], hip.ptr(ws), ws.numel(), hip.ptr(out), hip.stream()))
        return out

    def encoder_layer(self, prefix, src, has_pos):
        lib = hip.load()
        src = src.float().contiguous()
        ws = self._layer_ws(src.shape[0], src.shape[0])
        out = torch.empty_like(src)
        hip.check(lib.tgtc_s2d_encoder_layer(self.handle, prefix.encode(), hip.ptr(src), src.shape[0], int(has_pos),
                                             hip.ptr(ws), ws.numel(), hip.ptr(out), hip.stream()))
        return out

    def decoder_layer(self, prefix, tgt, memory, query_pos):
        lib = hip.load()
        tgt, memory, query_pos = (t.float().contiguous() for t in (tgt, memory, query_pos))
        ws = self._layer_ws(tgt.shape[0], memory.shape[0])
        out = torch.empty_like(tgt)
        hip.check(lib.tgtc_s2d_decoder_layer(self.handle, prefix.encode(), hip.ptr(tgt), tgt.shape[0], hip.ptr(memory),
                                             memory.shape[0], hip.ptr(query_pos), hip.ptr(ws), ws.numel(), hip.ptr(out),
                                             hip.stream()))
        return out

    def cnn_decode(self, tokens, h, w):
        """tokens [h*w,512] -> image [1,3,8h,8w]"""
        lib = hip.load()
        tokens = tokens.float().contiguous()
        ws = self.workspace(lib.tgtc_s2d_decode_workspace_bytes(h, w))
        img = torch.empty(1, 3, 8 * h, 8 * w, device=tokens.device)
        hip.check(lib.tgtc_s2d_cnn_decode(self.handle, hip.ptr(tokens), h, w, hip.ptr(ws), ws.numel(), hip.ptr(img),
                                          hip.stream()))
        return img

    def vgg_encode(self, img):
        """img [1,3,H,W] -> [relu1_1, relu2_1, relu3_1, relu4_1] NCHW"""
        lib = hip.load()
        _, _, H, W = img.shape
        img = img.float().contiguous()
        ws = self.workspace(lib.tgtc_s2d_vgg_workspace_bytes(H, W))
        dims, h, w = [], H, W
        for c in (64, 128, 256, 512):
            dims.append((c, h, w))
            h, w = (h + 1) // 2, (w + 1) // 2
        outs = [torch.empty(1, c, hh, ww, device=img.device) for c, hh, ww in dims]
        hip.check(lib.tgtc_s2d_vgg_encode(self.handle, hip.ptr(img), H, W, hip.ptr(ws), ws.numel(),
                                          *[hip.ptr(o) for o in outs], hip.stream()))
        return outs


# ----------------------------------------------------------------------------------------- stateless helpers
def calc_mean_std(feat, eps=1e-5):
    """reference function.py:4-12 / Style_function.py:4-12.  feat [N,C,H,W] -> (mean, std) [N,C,1,1]."""
    hip.require_gpu(feat)
    lib = hip.load()
    N, C = feat.shape[:2]
    f = feat.float().contiguous()
    hw = f[0, 0].numel()
    mean = torch.empty(N * C, device=f.device)
    std = torch.empty(N * C, device=f.device)
    hip.check(lib.tgtc_s2d_mean_std(hip.ptr(f), N * C, hw, float(eps), hip.ptr(mean), hip.ptr(std), hip.stream()))
    return mean.view(N, C, 1, 1), std.view(N, C, 1, 1)


def adaptive_instance_normalization(content_feat, style_feat):
    """reference Style_function.py:15-24 (batch of 1)."""
    hip.require_gpu(content_feat, style_feat)
    lib = hip.load()
    assert content_feat.shape[:2] == style_feat.shape[:2] and content_feat.shape[0] == 1
    C = content_feat.shape[1]
    c, s = content_feat.float().contiguous(), style_feat.float().contiguous()
    stats = torch.empty(4 * C, device=c.device)
    out = torch.empty_like(c)
    hip.check(lib.tgtc_s2d_adain(hip.ptr(c), c[0, 0].numel(), hip.ptr(s), s[0, 0].numel(), C, hip.ptr(stats),
                                 hip.ptr(out), hip.stream()))
    return out


def linear(x, weight, bias=None, relu=False, precision="fp16x3"):
    """nn.Linear (+ ReLU) on the HIP GEMM kernel: x [M,K], weight [N,K], bias [N] CUDA float32 -> [M,N]."""
    hip.require_gpu(x, weight)
    lib = hip.load()
    x, weight = x.float().contiguous(), weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K
    y = torch.empty(M, N, device=x.device)
    hip.check(lib.tgtc_s2d_linear(hip.ptr(x), M, K, hip.ptr(weight), hip.ptr(b), N, int(relu), hip.PRECISIONS[precision],
                                  hip.ptr(y), hip.stream()))
    return y


def resize_bilinear(img, size):
    """nn.Upsample(size, mode='bilinear', align_corners=True) (trans_test.py:172-173); img [1,C,h,w]."""
    hip.require_gpu(img)
    lib = hip.load()
    _, C, h, w = img.shape
    x = img.float().contiguous()
    out = torch.empty(1, C, size[0], size[1], device=x.device)
    hip.check(lib.tgtc_s2d_resize_bilinear(hip.ptr(x), C, h, w, hip.ptr(out), size[0], size[1], hip.stream()))
    return out


def style_feature(hs_tokens):
    """trans_test.py:176 on token-major hs [n,512] -> [1,1024] (the reference's reshape(-1,512) statistic)."""
    hip.require_gpu(hs_tokens)
    lib = hip.load()
    x = hs_tokens.float().contiguous()
    out = torch.empty(1024, device=x.device)
    hip.check(lib.tgtc_s2d_style_feature(hip.ptr(x), x.shape[0], hip.ptr(out), hip.stream()))
    return out[None]


def tokens_to_nchw(tokens, h, w):
    hip.require_gpu(tokens)
    lib = hip.load()
    n, C = tokens.shape
    t = tokens.float().contiguous()
    out = torch.empty(1, C, h, w, device=t.device)
    hip.check(lib.tgtc_s2d_tokens_to_nchw(hip.ptr(t), n, C, hip.ptr(out), hip.stream()))
    return out


def nchw_to_tokens(x):
    hip.require_gpu(x)
    lib = hip.load()
    _, C, h, w = x.shape
    t = x.float().contiguous()
    out = torch.empty(h * w, C, device=t.device)
    hip.check(lib.tgtc_s2d_nchw_to_tokens(hip.ptr(t), h * w, C, hip.ptr(out), hip.stream()))
    return out


# ----------------------------------------------------------------------------------------- reference-shaped modules
class _Lazy(nn.Module):
    """nn.Module whose parameters are mirrored into a Handle, rebuilt when they change."""
    precision = "fp16x3"

    def _key(self):
        return (self.precision,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def handle(self):
        key = self._key()
        if getattr(self, "_h", None) is None or self._hkey != key:
            self._h, self._hkey = self._make_handle(), key
        return self._h


class PatchEmbed(_Lazy):
    """reference tctrans.py:13-33."""

    def __init__(self, img_size=256, patch_size=8, in_chans=3, embed_dim=512):
        super().__init__()
        assert patch_size == 8 and in_chans == 3 and embed_dim == 512, "HIP kernel implements the 8x8, 3->512 embedding"
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def _make_handle(self):
        return Handle(embedding=self.state_dict(), precision=self.precision)

    def forward(self, x):
        _, _, H, W = x.shape
        return tokens_to_nchw(self.handle().patch_embed(x), H // 8, W // 8)


def _transformer_skeleton(d=512, ff=2048, n_enc=3, n_dec=3):
    """Parameter containers with the reference's names (transformer.py:13-44, :145-165, :209-228)."""
    def attn():
        m = nn.Module()
        m.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        m.in_proj_bias = nn.Parameter(torch.empty(3 * d))
        m.out_proj = nn.Linear(d, d)
        return m

    def enc_layer():
        m = nn.Module()
        m.qk, m.qkv = nn.Linear(d, 2 * d, bias=False), nn.Linear(d, 3 * d, bias=False)
        m.self_attn = attn()
        m.linear1, m.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        m.norm1, m.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)
        return m

    def dec_layer():
        m = nn.Module()
        m.self_attn, m.multihead_attn = attn(), attn()
        m.linear1, m.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        m.norm1, m.norm2, m.norm3 = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)
        return m

    def stack(make, n):
        m = nn.Module()
        m.layers = nn.ModuleList([make() for _ in range(n)])
        return m

    return stack(enc_layer, n_enc), stack(enc_layer, n_enc), stack(dec_layer, n_dec)


class Transformer(_Lazy):
    """reference transformer.py:13-75 (post-norm, ReLU, eval)."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=3, num_decoder_layers=3, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        assert (d_model, nhead, dim_feedforward, activation, normalize_before) == (512, 8, 2048, "relu", False), \
            "HIP kernels implement d=512, 8 heads, FFN 2048, post-norm ReLU"
        self.encoder_c, self.encoder_s, self.decoder = _transformer_skeleton(d_model, dim_feedforward,
                                                                             num_encoder_layers, num_decoder_layers)
        self.decoder.norm = nn.LayerNorm(d_model)
        self.new_ps = nn.Conv2d(512, 512, (1, 1))   # allocated but unused by the reference (transformer.py:38,51-53)
        self.d_model, self.nhead = d_model, nhead
        for p in self.parameters():     # transformer.py:41-44
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, nn.Module) and hasattr(m, "in_proj_bias"):
                nn.init.zeros_(m.in_proj_bias)

    def _make_handle(self):
        return Handle(transformer={k: v for k, v in self.state_dict().items() if not k.startswith("new_ps")},
                      precision=self.precision)

    def forward(self, style, mask, content, pos_embed_c, pos_embed_s):
        """style, content [1,512,h,w] patch embeddings.  As in the reference, `pos_embed_c` is only a switch for the
        content encoder and must be the content embedding (it is what the decoder adds to its queries)."""
        assert mask is None and pos_embed_s is None and pos_embed_c is not None
        if pos_embed_c is not content and not torch.equal(pos_embed_c, content):
            raise NotImplementedError("the reference only ever passes pos_embed_c = content (tctrans.py:238)")
        _, C, hs_, ws_ = style.shape
        hs = self.handle().transformer(nchw_to_tokens(style), nchw_to_tokens(content))
        return tokens_to_nchw(hs, hs_, ws_)     # reshaped with the STYLE map size (transformer.py:49,73)


class _Seq(_Lazy):
    """Parameter container with nn.Sequential-style integer names."""

    def __init__(self, convs):
        super().__init__()
        for idx, (o, i, k) in convs.items():
            self.add_module(str(idx), nn.Conv2d(i, o, (k, k)))


class Decoder(_Seq):
    """reference tctrans.py:36-66 `decoder`."""

    def __init__(self):
        from .synth import DECODER_SHAPES
        super().__init__({k: (o, i, 3) for k, (o, i) in DECODER_SHAPES.items()})

    def _make_handle(self):
        return Handle(decoder=self.state_dict(), precision=self.precision)

    def forward(self, x):
        _, _, h, w = x.shape
        return self.handle().cnn_decode(nchw_to_tokens(x), h, w)


class VGG(_Seq):
    """reference tctrans.py:68-99 `vgg` truncated to [:31] (all call sites slice it, trans_test.py:97)."""

    def __init__(self):
        from .synth import VGG_SHAPES
        super().__init__(VGG_SHAPES)

    def _make_handle(self):
        return Handle(vgg=self.state_dict(), precision=self.precision)

    def encode_with_intermediate(self, x):
        f = self.handle().vgg_encode(x)
        return f + [f[-1]]          # enc_5 is empty for vgg[:31] -> identity (tctrans.py:146,161-166)


class StyTrans(nn.Module):
    """reference tctrans.py:138-245, forward = the TEST branch (:233-245).  The reference picks the training branch
    whenever H == W (:187) and then returns five values where trans_test.py:164 unpacks two; the north-star frame is
    square, so this class always runs the test-branch computation."""

    def __init__(self, encoder, decoder, PatchEmbed, transformer):
        super().__init__()
        self.encoder, self.decode, self.embedding, self.transformer = encoder, decoder, PatchEmbed, transformer

    def encode_with_intermediate(self, x):
        return self.encoder.encode_with_intermediate(x)

    def forward(self, samples_c, samples_s):
        style = self.embedding(samples_s)
        content = self.embedding(samples_c)
        hs = self.transformer(style, None, content, content, None)
        return self.decode(hs), hs


def stylize_frame(net, content, style):
    """What trans_test.transformer_render does per frame (trans_test.py:164-176): returns the stylised image resized
    to the content size and the 1024-d style feature row."""
    ics, hs = net(content, style)
    image = resize_bilinear(ics, content.shape[-2:])
    return image, style_feature(nchw_to_tokens(hs)), hs


def stylize_frames(net, contents, style, output_path=None, style_name="style", style_path="", style_image=None, save_ext=".png"):
    """The frame loop of trans_test.transformer_render (trans_test.py:151-179) for one style image.

    contents: iterable of [1,3,h,w] CUDA tensors in [0,1]; style: [1,3,h,w].  Per frame: network, bilinear resize to the
    content size (align_corners=True, :172-173), image file `NNN<ext>` counted from 001 (:168-170; torchvision.save_image
    = x*255+0.5 clamped to uint8), and one row [mean_tokens(hs), var_tokens(hs)] (:176).  The rows are averaged exactly as
    the reference does -- a zero row is prepended and the sum divided by rows-1 (:145,178) -- and written with the style
    image to `stylized_data.npz` under the reference's keys (:179), which is what its dataset reads back
    (dataset.py:437-440).  Returns (list of uint8 HWC images, style_features [1,1024] float32)."""
    from .image_writer import writer
    feats, images = [], []
    if output_path is not None:
        os.makedirs(output_path, exist_ok=True)
    for cnt, content in enumerate(contents, start=1):
        image, feat, _ = stylize_frame(net, content, style)
        feats.append(feat.detach().float().reshape(1024))         # stays on the device until the loop is done
        img8 = image[0].detach().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).contiguous()
        images.append(img8)
        if output_path is not None:                                  # encoded and written in the background
            writer().save('{:s}/{:03d}{:s}'.format(output_path, cnt, save_ext), img8)
    writer().drain()
    images = [im.cpu().numpy() for im in images]
    rows = np.zeros([1, 1024], dtype=np.float32)
    if feats:
        rows = np.append(rows, torch.stack(feats).cpu().numpy(), axis=0)
    features = np.sum(rows, axis=0, keepdims=True) / (rows.shape[0] - 1)
    if output_path is not None:
        if style_image is None:
            style_image = np.moveaxis(style.detach().float().cpu().numpy(), 1, -1)          # [1,h,w,3] like :144
        np.savez(os.path.join(output_path, 'stylized_data'), style_names={style_name: 0}, style_paths=style_path,
                 style_images=style_image, style_features=features)
    return images, features
