"""Oracle: forward pass of the 2-D style module (PyTorch CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference tctrans.py:13-33 (PatchEmbed), :36-66 (CNN decoder), :68-99 + :161-166
(VGG encoder up to relu4_1, `encode_with_intermediate`), :233-245 (StyTrans test branch),
transformer.py:46-75 (Transformer.forward), :167-184 (encoder layer, post-norm),
:236-263 (decoder layer, post-norm), function.py:4-12 (calc_mean_std) and
trans_test.py:172-179 (bilinear resize + 1024-d style feature).

All networks are evaluated functionally from state dicts with the reference's key names.
"""
import math

import torch
import torch.nn.functional as F

D_MODEL = 512
N_HEAD = 8


def mha(sd, prefix, query, key, value):
    """nn.MultiheadAttention forward, batch of 1 folded away: inputs are [L, 512] / [S, 512].

    Packed in_proj (rows 0-511 q, 512-1023 k, 1024-1535 v) with bias, 8 heads of 64, scores
    scaled by 1/sqrt(64), softmax over keys, out_proj.  (transformer.py:158,177 use the torch module.)
    """
    w, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    q = F.linear(query, w[:D_MODEL], b[:D_MODEL])
    k = F.linear(key, w[D_MODEL:2 * D_MODEL], b[D_MODEL:2 * D_MODEL])
    v = F.linear(value, w[2 * D_MODEL:], b[2 * D_MODEL:])
    L, S, dh = q.shape[0], k.shape[0], D_MODEL // N_HEAD
    q = q.view(L, N_HEAD, dh).transpose(0, 1) / math.sqrt(dh)
    k = k.view(S, N_HEAD, dh).transpose(0, 1)
    v = v.view(S, N_HEAD, dh).transpose(0, 1)
    p = torch.softmax(q @ k.transpose(1, 2), -1)
    o = (p @ v).transpose(0, 1).reshape(L, D_MODEL)
    return F.linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def _ln(sd, name, x):
    return F.layer_norm(x, (D_MODEL,), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _ffn(sd, p, x):
    h = torch.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"]))
    return F.linear(h, sd[p + "linear2.weight"], sd[p + "linear2.bias"])


def encoder_layer(sd, p, src, has_pos):
    """TransformerEncoderLayer.forward_post.  transformer.py:167-184.

    pos is only a switch (:172-176): without it q,k and the *value/residual* come from the qkv
    projection; with it q,k come from the qk projection and value/residual stay src.
    """
    if not has_pos:
        q, k, src = F.linear(src, sd[p + "qkv.weight"]).chunk(3, -1)
    else:
        q, k = F.linear(src, sd[p + "qk.weight"]).chunk(2, -1)
    src = _ln(sd, p + "norm1", src + mha(sd, p + "self_attn.", q, k, src))
    return _ln(sd, p + "norm2", src + _ffn(sd, p, src))


def decoder_layer(sd, p, tgt, memory, query_pos):
    """TransformerDecoderLayer.forward_post with pos=None.  transformer.py:236-263.
    Both attention blocks attend tgt+query_pos -> memory."""
    tgt = _ln(sd, p + "norm1", tgt + mha(sd, p + "self_attn.", tgt + query_pos, memory, memory))
    tgt = _ln(sd, p + "norm2", tgt + mha(sd, p + "multihead_attn.", tgt + query_pos, memory, memory))
    return _ln(sd, p + "norm3", tgt + _ffn(sd, p, tgt))


def transformer_forward(sd, style_map, content_map, n_enc=3, n_dec=3):
    """Transformer.forward(style, None, content, pos_c=content, pos_s=None).  transformer.py:46-75.

    style_map / content_map: [1, 512, h, w] patch embeddings.  Returns hs [1, 512, hs, ws]
    (the reshape uses the *style* map's spatial size, :49,:73).
    """
    _, C, hs_, ws_ = style_map.shape
    s = style_map.flatten(2)[0].t()
    c = content_map.flatten(2)[0].t()
    qpos = c
    for i in range(n_enc):
        s = encoder_layer(sd, "encoder_s.layers.%d." % i, s, has_pos=False)
    for i in range(n_enc):
        c = encoder_layer(sd, "encoder_c.layers.%d." % i, c, has_pos=True)
    out = c
    for i in range(n_dec):
        out = decoder_layer(sd, "decoder.layers.%d." % i, out, s, qpos)
    out = _ln(sd, "decoder.norm", out)
    return out.t().reshape(1, C, hs_, ws_)


def patch_embed(sd, img, patch=8):
    """PatchEmbed.forward: Conv2d(3,512,k=8,s=8).  tctrans.py:26,29-33."""
    return F.conv2d(img, sd["proj.weight"], sd["proj.bias"], stride=patch)


# (in_ch, out_ch, relu, upsample_after) per 3x3 conv of the CNN decoder, tctrans.py:36-66;
# values are the Sequential indices of the conv layers.
DECODER_CONVS = [(1, True, True), (5, True, False), (8, True, False), (11, True, False), (14, True, True),
                 (18, True, False), (21, True, True), (25, True, False), (28, False, False)]


def cnn_decode(sd, x):
    """The CNN decoder: 9 x (ReflectionPad(1) + Conv3x3), ReLU after all but the last, nearest x2
    upsample after convs 1, 5 and 7.  tctrans.py:36-66."""
    for idx, relu, up in DECODER_CONVS:
        x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd["%d.weight" % idx], sd["%d.bias" % idx])
        if relu:
            x = torch.relu(x)
        if up:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
    return x


# VGG-19 prefix up to relu4_1, tctrans.py:68-99: ('c1', idx) 1x1 conv, ('c', idx) pad+3x3 conv+relu,
# ('p',) 2x2 ceil-mode max-pool, ('tap',) marks the returned activations (enc_1..enc_4, tctrans.py:143-146).
VGG_PLAN = [("c1", 0), ("c", 2), ("tap",), ("c", 5), ("p",), ("c", 9), ("tap",), ("c", 12), ("p",),
            ("c", 16), ("tap",), ("c", 19), ("c", 22), ("c", 25), ("p",), ("c", 29), ("tap",)]


def vgg_encode(sd, img):
    """StyTrans.encode_with_intermediate over vgg[:31].  tctrans.py:161-166.
    Returns [relu1_1, relu2_1, relu3_1, relu4_1, relu4_1] (enc_5 is empty -> identity)."""
    x, taps = img, []
    for step in VGG_PLAN:
        if step[0] == "c1":
            x = F.conv2d(x, sd["%d.weight" % step[1]], sd["%d.bias" % step[1]])
        elif step[0] == "c":
            x = torch.relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"),
                                    sd["%d.weight" % step[1]], sd["%d.bias" % step[1]]))
        elif step[0] == "p":
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
        else:
            taps.append(x)
    return taps + [taps[-1]]


def mean_std(feat, eps=1e-5):
    """calc_mean_std: per-(N,C) mean and sqrt(unbiased var + eps).  function.py:4-12."""
    n, c = feat.shape[:2]
    flat = feat.reshape(n, c, -1)
    return flat.mean(2).view(n, c, 1, 1), (flat.var(2) + eps).sqrt().view(n, c, 1, 1)


def adain(content, style):
    """adaptive_instance_normalization.  Style_function.py:15-24."""
    sm, ss = mean_std(style)
    cm, cs = mean_std(content)
    return (content - cm) / cs * ss + sm


def style_feature(hs):
    """trans_test.py:176 (see the note in `stylize`).  hs [1,512,h,w] -> [1,1024]."""
    rows = hs.reshape(-1, D_MODEL)
    return torch.cat([rows.mean(0), rows.var(0)])[None]


def stylize(sd_embed, sd_trans, sd_dec, content, style):
    """StyTrans test branch + the post-processing of trans_test.py.

    tctrans.py:233-245: embed both images, transformer, CNN decoder.
    trans_test.py:172-173: bilinear resize of the output to the content size, align_corners=True.
    trans_test.py:176: style feature = [rows.mean(0), rows.var(0)] with rows = hs.reshape(-1, 512).
    NOTE hs is [1,512,h,w]; the reference reshapes the (c,h,w)-ordered flattening into rows of 512
    consecutive values, so unless h*w == 512 a "row" is NOT one token: column j collects the elements
    whose flat index is j mod 512.  The oracle reproduces that exactly (unbiased variance).
    """
    s = patch_embed(sd_embed, style)
    c = patch_embed(sd_embed, content)
    hs = transformer_forward(sd_trans, s, c)
    ics = cnn_decode(sd_dec, hs)
    out = F.interpolate(ics, size=content.shape[-2:], mode="bilinear", align_corners=True)
    feat = style_feature(hs)
    return {"hs": hs, "ics": ics, "image": out, "style_feature": feat}
