"""Neural operators with the reference's `models.py` class names, constructor arguments, forward
keyword signatures, returned dict keys and state-dict key names -- evaluated by the HIP library.

reference: models.py:24-60 (Embedder), :63-117 (MLP_style), :120-147 (StyleMLP_before_concat),
:149-180 (StyleMLP_Wild_multilayers), :182-223 (StyleNerf), :475-506 (StyleLatents_variational).

The modules own ordinary `nn.Parameter`s under the reference's names (so reference checkpoints
`load_state_dict` directly); the packed fp16 weight stream the kernels read is (re)built lazily
whenever the parameters change.  The render paths never backpropagate and run the fused kernels; when gradients are
enabled on a network marked `.trainable()` (the reference's training loops) StyleNerf / MLP_style switch to differentiable
layer-by-layer HIP dense layers (autograd_ops.py).
Select the arithmetic mode with `args.precision` = 'fp16x3' (default, fp32-equivalent) or 'fp16'.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import hip


def _precision_of(args, role):
    """`--precision` is one mode for every network or "coarse+fine" (precision belongs to each packed handle): the
    coarse NeRF takes the first, the fine NeRF the second; the style MLPs are built for fp16x3 / fp16 and fall back
    to fp16x3 where the fine NeRF runs fp16mx."""
    p = getattr(args, 'precision', 'fp16x3')
    first, _, second = p.partition('+')
    second = second or first
    if role == 'coarse':
        return first
    if role == 'fine':
        return second
    return second if second in ('fp16x3', 'fp16') else 'fp16x3'


class Embedder(nn.Module):
    """reference models.py:24-60 (log-sampled bands, include_input, sin/cos)."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True):
        super().__init__()
        if not (log_sampling and include_input and input_dim == 3 and max_freq_log2 == N_freqs - 1):
            raise NotImplementedError("the HIP encoder implements the reference's only configuration: "
                                      "3-d input, bands 2^0..2^(L-1), include_input")
        self.input_dim, self.N_freqs = input_dim, N_freqs
        self.out_dim = input_dim * (1 + 2 * N_freqs)

    def forward(self, x):
        hip.require_gpu(x)
        lib = hip.load()
        lead = x.shape[:-1]
        is64 = x.dtype == torch.float64
        flat = x.reshape(-1, 3).to(torch.float64 if is64 else torch.float32).contiguous()
        out = torch.empty(flat.shape[0], self.out_dim, device=x.device, dtype=torch.float32)
        hip.check(lib.tgtc_posenc(hip.ptr(flat), int(is64), flat.shape[0], self.N_freqs, hip.ptr(out), hip.stream()))
        # the reference returns the input dtype and StyleNerf casts to float32 (models.py:219-220)
        return out.reshape(*lead, self.out_dim)


class _Packed(nn.Module):
    """Caches a device-resident packed copy of the parameters; invalidated when they change."""

    def __init__(self):
        super().__init__()
        self._net = None
        self._net_key = None
        self.precision = "fp16x3"

    def _param_key(self):
        return (self.precision,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _packed(self):
        key = self._param_key()
        if self._net is None or key != self._net_key:
            self._net = self._pack()
            self._net_key = key
        return self._net


class MLP_style(_Packed):
    """reference models.py:63-117 (D=8, W=256, skips=[4], ReLU, view-dependent colour)."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_viewdirs=27, skips=(4,), act_func=nn.ReLU, use_viewdir=True,
                 sigma_mul=0., enable_style=False):
        super().__init__()
        self.input_ch, self.input_ch_viewdirs, self.skips = input_ch, input_ch_viewdirs, list(skips)
        self.use_viewdir, self.sigma_mul, self.enable_style = use_viewdir, sigma_mul, enable_style
        layers, dim = [], input_ch
        for i in range(D):
            layers.append(nn.Linear(dim, W))
            dim = W
            if i in self.skips and i != D - 1:
                dim += input_ch
        self.base_layers = nn.ModuleList(layers)
        self.sigma_layer = nn.Linear(dim, 1)
        self.base_remap_layer = nn.Linear(dim, 256)
        d = 256 + input_ch_viewdirs if use_viewdir else 256
        self.rgb_layers = nn.ModuleList([nn.Linear(d, W // 2), nn.Linear(W // 2, 3)])
        if act_func is not nn.ReLU or sigma_mul != 0.:
            raise NotImplementedError("HIP kernels implement the ReLU network (act_type=relu)")

    def _pack(self):
        return hip.nerf_create(self.state_dict(), self.precision, prefix="")

    differentiable = False      # set True (StyleNerf.trainable()) to get gradients: the default is the fused forward-only path

    def wants_grad(self):
        return self.differentiable and torch.is_grad_enabled()

    def forward_layers(self, pe, de):
        """The same network layer by layer on differentiable HIP dense layers (autograd_ops.py): reference
        models.py:95-111.  pe [M,63], de [M,27] float32 -> (rgb [M,3], base_remap [M,256], sigma [M])."""
        from . import autograd_ops as ao
        prec = self.precision if self.precision in ("fp16x3", "fp16") else "fp16x3"
        base = ao.linear(pe, self.base_layers[0], True, prec)
        for i in range(len(self.base_layers) - 1):
            if i in self.skips:
                base = torch.cat((pe, base), dim=-1)
            base = ao.linear(base, self.base_layers[i + 1], True, prec)
        sigma = ao.linear(base, self.sigma_layer, False, prec)
        remap = ao.linear(base, self.base_remap_layer, True, prec)
        fea = ao.linear(torch.cat((remap, de), dim=-1) if self.use_viewdir else remap, self.rgb_layers[0], True, prec)
        rgb = ao.sigmoid(ao.linear(fea, self.rgb_layers[1], False, prec))
        return rgb, remap, sigma.squeeze(-1)

    def forward(self, **kwargs):
        """pts [..,63], dirs [..,27] float32 (already encoded) -> dict rgb, base_remap, pts, sigma."""
        pts, dirs = kwargs['pts'], kwargs['dirs']
        hip.require_gpu(pts, dirs)
        lib = hip.load()
        lead = pts.shape[:-1]
        if self.wants_grad():
            rgb, remap, sigma = self.forward_layers(pts.reshape(-1, self.input_ch).to(torch.float32).contiguous(),
                                                    dirs.reshape(-1, self.input_ch_viewdirs).to(torch.float32).contiguous())
            return OrderedDict([('rgb', rgb.reshape(*lead, 3)), ('base_remap', remap.reshape(*lead, 256)),
                                ('pts', pts), ('sigma', sigma.reshape(*lead))])
        p = pts.reshape(-1, self.input_ch).to(torch.float32).contiguous()
        d = dirs.reshape(-1, self.input_ch_viewdirs).to(torch.float32).contiguous()
        M = p.shape[0]
        rgb = torch.empty(M, 3, device=p.device, dtype=torch.float32)
        sigma = torch.empty(M, device=p.device, dtype=torch.float32)
        remap = torch.empty(M, 256, device=p.device, dtype=torch.float32)
        hip.check(lib.tgtc_nerf_mlp_forward(self._packed().handle, hip.ptr(p), hip.ptr(d), M, hip.ptr(rgb),
                                            hip.ptr(sigma), hip.ptr(remap), hip.stream()))
        return OrderedDict([('rgb', rgb.reshape(*lead, 3)), ('base_remap', remap.reshape(*lead, 256)),
                            ('pts', pts), ('sigma', sigma.reshape(*lead))])


class StyleNerf(nn.Module):
    """reference models.py:182-223."""

    def __init__(self, args, mode='coarse', enable_style=False):
        super().__init__()
        self.use_viewdir = args.use_viewdir
        if getattr(args, 'act_type', 'relu') != 'relu':
            raise NotImplementedError("HIP kernels implement act_type=relu (the shipped configs)")
        self.is_siren = False
        self.embedder_coor = Embedder(3, args.embed_freq_coor - 1, args.embed_freq_coor)
        self.embedder_dir = Embedder(3, args.embed_freq_dir - 1, args.embed_freq_dir)
        depth, width = ((args.netdepth, args.netwidth) if mode == 'coarse'
                        else (args.netdepth_fine, args.netwidth_fine))
        self.net = MLP_style(D=depth, W=width, input_ch=self.embedder_coor.out_dim,
                             input_ch_viewdirs=self.embedder_dir.out_dim, skips=[4], use_viewdir=self.use_viewdir,
                             enable_style=enable_style)
        self.net.precision = _precision_of(args, mode)
        self.enable_style = enable_style
        self.fused_training, self._trainer = False, None

    def trainable(self, on=True, fused=True):
        """Training side: whenever gradients are enabled the forward becomes differentiable w.r.t. the 24 parameters (the
        render paths stay on the fused forward-only kernels).  fused=True (default): the fused training path
        (fused_train.py / csrc/mlp_train.hip: forward with activation stash, input-gradient chain and weight gradients as
        three launches; returns the dict entries a training body reads -- rgb and sigma).  fused=False: the same arithmetic
        layer by layer on differentiable HIP dense layers (autograd_ops.py), with every entry of the render-path dict."""
        self.net.differentiable = bool(on)
        self.fused_training = bool(on and fused)
        if self.fused_training and self._trainer is None:
            from . import fused_train
            self._trainer = fused_train.NerfTrainer()
        if not on and self._trainer is not None:
            self._trainer.drop_workspaces()         # ~20 KB per sample of the largest batch seen (fused_train.NerfTrainer)
        return self

    def training_overflows(self):
        """Backwards of the fused training path whose gradients were zero-filled by the library's overflow guard (the scaled
        input gradients left the fp16 range; include/tgtc_train.h).  Synchronises: read it where the reference prints its
        losses (every i_print iterations, train_tgtcs.py:257-266)."""
        return 0 if self._trainer is None else self._trainer.overflows()

    def set_enable_style(self, enable_style=False):
        self.enable_style = enable_style
        self.net.enable_style = enable_style

    def packed(self):
        return self.net._packed()

    def forward(self, **kwargs):
        """pts, dirs [..,3] (float64 in the render path) -> dict rgb, base_remap, pts (encoded), sigma, dirs (encoded)."""
        pts, dirs = kwargs['pts'], kwargs['dirs']
        hip.require_gpu(pts)
        lib = hip.load()
        lead = pts.shape[:-1]
        if self.net.wants_grad() and self.fused_training:
            rgb, sigma = self._trainer.apply(self.net, pts.detach(), dirs.detach())      # train_tgtcs.py:234-236 reads rgb, sigma
            return OrderedDict([('rgb', rgb), ('sigma', sigma)])
        if self.net.wants_grad():
            # training side (train_tgtcs.py:218-309): encodings from the HIP encoder (no gradient: the samplers are detached
            # in the reference too), then the differentiable layer-by-layer network
            pe = self.embedder_coor(pts.reshape(-1, 3).detach())
            de = self.embedder_dir(dirs.expand(*lead, 3).reshape(-1, 3).detach())
            rgb, remap, sigma = self.net.forward_layers(pe, de)
            return OrderedDict([('rgb', rgb.reshape(*lead, 3)), ('base_remap', remap.reshape(*lead, 256)),
                                ('pts', pe.reshape(*lead, self.embedder_coor.out_dim)), ('sigma', sigma.reshape(*lead)),
                                ('dirs', de.reshape(*lead, self.embedder_dir.out_dim))])
        p = pts.reshape(-1, 3).to(torch.float64).contiguous()
        d = dirs.expand(*lead, 3).reshape(-1, 3).to(torch.float64).contiguous()
        M = p.shape[0]
        dev = p.device
        rgb = torch.empty(M, 3, device=dev, dtype=torch.float32)
        sigma = torch.empty(M, device=dev, dtype=torch.float32)
        remap = torch.empty(M, 256, device=dev, dtype=torch.float32)
        pe = torch.empty(M, self.embedder_coor.out_dim, device=dev, dtype=torch.float32)
        de = torch.empty(M, self.embedder_dir.out_dim, device=dev, dtype=torch.float32)
        hip.check(lib.tgtc_nerf_forward(self.packed().handle, hip.ptr(p), hip.ptr(d), M, hip.ptr(rgb), hip.ptr(sigma),
                                        hip.ptr(remap), hip.ptr(pe), hip.ptr(de), hip.stream()))
        return OrderedDict([('rgb', rgb.reshape(*lead, 3)), ('base_remap', remap.reshape(*lead, 256)),
                            ('pts', pe.reshape(*lead, self.embedder_coor.out_dim)), ('sigma', sigma.reshape(*lead)),
                            ('dirs', de.reshape(*lead, self.embedder_dir.out_dim))])


class StyleMLP_before_concat(_Packed):
    """reference models.py:120-147: 5 linears (the constructor loop breaks at the skip), every layer on
    cat(h, latent) and the last additionally on x."""

    def __init__(self, args):
        super().__init__()
        pe = args.embed_freq_coor * 3 * 2 + 3
        # any shape can be built, loaded and saved (checkpoint files); the kernels behind forward() have one
        self._unsupported = (args.style_D != 8 or args.netwidth != 256 or pe != 63 or args.vae_latent != 32)
        self.skips = [4]
        dims, dim = [], pe + args.vae_latent
        for i in range(args.style_D - 1):
            if i in self.skips:
                dims.append(dim + pe)
                break
            dims.append(dim)
            dim = args.netwidth + args.vae_latent
        self.layers = nn.ModuleList([nn.Linear(d, args.netwidth) for d in dims])
        self.precision = _precision_of(args, 'style')

    def _require_supported(self):
        if self._unsupported:
            raise NotImplementedError("HIP kernels implement style_D=8, netwidth=256, embed_freq_coor=10, vae_latent=32")

    def _pack(self):
        self._require_supported()
        return hip.style_create(concat_state=self.state_dict(), precision=self.precision)

    differentiable = False
    act_trace = None            # tests: a list that receives every hidden activation of the differentiable forward

    def trainable(self, on=True):
        """Training side (Style_train, train_tgtcs.py:312-571): forward on the differentiable HIP dense layers."""
        self.differentiable = bool(on)
        return self

    def forward(self, **kwargs):
        x, latent = kwargs['x'], kwargs['latent']
        self._require_supported()
        hip.require_gpu(x, latent)
        lib = hip.load()
        lead = x.shape[:-1]
        xf = x.reshape(-1, 63).to(torch.float32).contiguous()
        lf = latent.expand(*lead, 32).reshape(-1, 32).to(torch.float32).contiguous()
        if self.differentiable and torch.is_grad_enabled():       # models.py:137-147, layer by layer
            from . import autograd_ops as ao
            h = xf
            for i, layer in enumerate(self.layers):
                h = torch.cat([h, lf], -1)
                if i in self.skips:
                    h = torch.cat([h, xf], -1)
                h = ao.linear(h, layer, True, self.precision)
                if self.act_trace is not None:
                    self.act_trace.append(h.detach())
            return {'concat_features': h.reshape(*lead, 256)}
        out = torch.empty(xf.shape[0], 256, device=xf.device, dtype=torch.float32)
        hip.check(lib.tgtc_concat_mlp_forward(self._packed().handle, hip.ptr(xf), hip.ptr(lf), xf.shape[0],
                                              hip.ptr(out), hip.stream()))
        return {'concat_features': out.reshape(*lead, 256)}


class StyleMLP_Wild_multilayers(_Packed):
    """reference models.py:149-180: 7 hidden linears + a 3-wide sigmoid head."""

    def __init__(self, args):
        super().__init__()
        pe = args.embed_freq_coor * 3 * 2 + 3
        # any shape can be built, loaded and saved (checkpoint files); the kernels behind forward() have one
        self._unsupported = (args.style_D != 8 or args.netwidth != 256 or pe != 63 or args.vae_latent != 32)
        self.skips = [4]
        dims, dim = [], pe + 512 + args.vae_latent
        for i in range(args.style_D - 1):
            if i in self.skips:
                dim += pe
            dims.append(dim)
            dim = args.netwidth + args.vae_latent
        self.layers = nn.ModuleList([nn.Linear(d, args.netwidth) for d in dims] +
                                    [nn.Linear(args.netwidth + args.vae_latent, 3)])
        self.precision = _precision_of(args, 'style')

    def _require_supported(self):
        if self._unsupported:
            raise NotImplementedError("HIP kernels implement style_D=8, netwidth=256, embed_freq_coor=10, vae_latent=32")

    def _pack(self):
        self._require_supported()
        return hip.style_create(style_state=self.state_dict(), precision=self.precision)

    differentiable = False
    act_trace = None            # tests: a list that receives every hidden activation of the differentiable forward

    def trainable(self, on=True):
        """Training side (Style_train, train_tgtcs.py:312-571): forward on the differentiable HIP dense layers."""
        self.differentiable = bool(on)
        return self

    def forward(self, **kwargs):
        x, conc, latent = kwargs['x'], kwargs['concated'], kwargs['latent']
        self._require_supported()
        hip.require_gpu(x, conc, latent)
        lib = hip.load()
        lead = x.shape[:-1]
        xf = x.reshape(-1, 63).to(torch.float32).contiguous()
        cf = conc.reshape(-1, 512).to(torch.float32).contiguous()
        lf = latent.expand(*lead, 32).reshape(-1, 32).to(torch.float32).contiguous()
        if self.differentiable and torch.is_grad_enabled():       # models.py:165-180, layer by layer
            from . import autograd_ops as ao
            h = torch.cat([cf, xf], -1)
            for i, layer in enumerate(self.layers[:-1]):
                h = torch.cat([h, lf], -1)
                if i in self.skips:
                    h = torch.cat([h, xf], -1)
                h = ao.linear(h, layer, True, self.precision)
                if self.act_trace is not None:
                    self.act_trace.append(h.detach())
            rgb = ao.sigmoid(ao.linear(torch.cat([h, lf], -1), self.layers[-1], False, self.precision))
            return {'rgb': rgb.reshape(*lead, 3)}
        out = torch.empty(xf.shape[0], 3, device=xf.device, dtype=torch.float32)
        hip.check(lib.tgtc_style_mlp_forward(self._packed().handle, hip.ptr(xf), hip.ptr(cf), hip.ptr(lf),
                                             xf.shape[0], hip.ptr(out), hip.stream()))
        return {'rgb': out.reshape(*lead, 3)}


class StylePair(_Packed):
    """The two style MLPs packed into ONE handle for the fused stylised kernel (rendering.RayRenderer)."""

    def __init__(self, concat_model, style_model, precision=None):
        super().__init__()
        self.concat_model, self.style_model = concat_model, style_model
        self.precision = precision or concat_model.precision

    def _pack(self):
        self.concat_model._require_supported(), self.style_model._require_supported()
        return hip.style_create(self.concat_model.state_dict(), self.style_model.state_dict(), self.precision)

    def packed(self):
        return self._packed()


class VAE(nn.Module):
    """reference models.py:371-457 as a parameter container with the same state-dict keys; only what the render path
    needs runs: `encode(x, various=False)` -> (z = mu, mu, log_var), the D-1 Linear+ReLU layers and the two heads of
    VAE_encoder.forward (:390-395) on the HIP GEMM kernel (tgtc_s2d_linear).  The decoder's parameters are carried so
    that the reference's vae.pth loads; sampling / losses belong to training."""

    def __init__(self, data_dim, latent_dim, W=512, D=4, kl_lambda=0.1, precision="fp16x3"):
        super().__init__()
        self.data_dim, self.latent_dim, self.W, self.D, self.kl_lambda, self.precision = data_dim, latent_dim, W, D, kl_lambda, precision

        def stack(first, out_name, out_dim):
            m = nn.Module()
            m.fc_layers = nn.ModuleList([nn.Linear(first if i == 0 else W, W) for i in range(D - 1)])
            for name, dim in zip(out_name, out_dim):
                setattr(m, name, nn.Linear(W, dim))
            return m
        self.encoder = stack(data_dim, ["fc_layer_mu", "fc_layer_log_var"], [latent_dim, latent_dim])
        self.decoder = stack(latent_dim, ["output_layer"], [data_dim])

    def encode(self, x, various=False):
        from . import style2d
        if various:
            raise NotImplementedError("VAE.encode(various=True) draws a training-time sample; the render path reads mu / log_var")
        h = x
        for layer in self.encoder.fc_layers:
            h = style2d.linear(h, layer.weight, layer.bias, relu=True, precision=self.precision)
        mu = style2d.linear(h, self.encoder.fc_layer_mu.weight, self.encoder.fc_layer_mu.bias, precision=self.precision)
        log_var = style2d.linear(h, self.encoder.fc_layer_log_var.weight, self.encoder.fc_layer_log_var.bias, precision=self.precision)
        return mu, mu, log_var


class _LatentGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, latents, mu, sid, fid, sigma_scale, tile7):
        lib = hip.load()
        S, F, D = latents.shape
        out = torch.empty(sid.shape[0], D, device=latents.device, dtype=torch.float32)
        hip.check(lib.tgtc_latents_forward(hip.ptr(latents.detach().float().contiguous()), hip.ptr(mu.detach().float().contiguous()), S, F, D,
                                           hip.ptr(sid), hip.ptr(fid), sid.shape[0], sigma_scale, int(tile7), hip.ptr(out), hip.stream()))
        ctx.save_for_backward(sid, fid)
        ctx.shape, ctx.sigma_scale, ctx.tile7 = (S, F, D), sigma_scale, tile7
        return out

    @staticmethod
    def backward(ctx, g):
        lib = hip.load()
        sid, fid = ctx.saved_tensors
        S, F, D = ctx.shape
        g = g.float().contiguous()
        d_lat = torch.zeros(S, F, D, device=g.device) if ctx.needs_input_grad[0] else None
        d_mu = torch.zeros(S, D, device=g.device) if ctx.needs_input_grad[1] else None
        hip.check(lib.tgtc_latents_backward(hip.ptr(g), S, F, D, hip.ptr(sid), hip.ptr(fid), sid.shape[0], ctx.sigma_scale,
                                            int(ctx.tile7), hip.ptr(d_lat), hip.ptr(d_mu), hip.stream()))
        return d_lat, d_mu, None, None, None, None


class StyleLatents_variational(nn.Module):
    """reference models.py:475-506, :535-539."""

    def __init__(self, **kwargs):
        super().__init__()
        self.style_num, self.frame_num, self.latent_dim = kwargs['style_num'], kwargs['frame_num'], kwargs['latent_dim']
        self.latents = nn.Parameter(torch.randn(self.style_num, self.frame_num, self.latent_dim))
        self.style_latents_mu = nn.Parameter(torch.randn(self.style_num, self.latent_dim))
        self.style_latents_logvar = nn.Parameter(torch.randn(self.style_num, self.latent_dim))
        self.sigma_scale = 1.

    def set_latents(self, generator=None):
        """models.py:535-539 with reparameterize (:421-424): every frame's latent = mu + exp(0.5 logvar) * N(0,1).
        Host-side initialisation of a parameter table (the draw is torch's generator, as in the reference)."""
        shape = [self.style_num, self.frame_num, self.latent_dim]
        mu = self.style_latents_mu.detach().unsqueeze(1).expand(shape)
        std = torch.exp(0.5 * self.style_latents_logvar.detach()).unsqueeze(1).expand(shape)
        eps = torch.randn(shape, generator=generator, device="cpu").to(mu.device)
        self.latents = nn.Parameter(eps * std + mu)

    def forward(self, **kwargs):
        style_ids, frame_ids = kwargs['style_ids'], kwargs['frame_ids']
        hip.require_gpu(self.latents)
        lib = hip.load()
        dev = self.latents.device
        sid = style_ids.to(device=dev, dtype=torch.int64).contiguous()
        fid = frame_ids.to(device=dev, dtype=torch.int64).contiguous()
        tile7 = kwargs.get('type', 'llff') == 'llff'
        rows = self.style_num * self.frame_num
        flat = sid * self.frame_num + fid
        limit = 7 * rows if tile7 else rows
        if flat.numel() and (int(flat.max()) >= limit or int(flat.min()) < 0):
            raise IndexError("latent index out of range (%d rows%s)" % (rows, ", tiled x7" if tile7 else ""))
        if torch.is_grad_enabled() and (self.latents.requires_grad or self.style_latents_mu.requires_grad) and self.differentiable:
            return _LatentGather.apply(self.latents, self.style_latents_mu, sid, fid, float(self.sigma_scale), bool(tile7))
        out = torch.empty(sid.shape[0], self.latent_dim, device=dev, dtype=torch.float32)
        hip.check(lib.tgtc_latents_forward(hip.ptr(self.latents.detach().contiguous()),
                                           hip.ptr(self.style_latents_mu.detach().contiguous()), self.style_num,
                                           self.frame_num, self.latent_dim, hip.ptr(sid), hip.ptr(fid), sid.shape[0],
                                           float(self.sigma_scale), int(tile7), hip.ptr(out), hip.stream()))
        return out

    def minus_logp(self, **kwargs):
        """models.py:526-533: sum((z - mu)^2 / (exp(0.5 logvar) + 1e-3)) over the latent, mean over the rays (mu and logvar
        detached).  Loss-side arithmetic on [R,32] tensors, as in the reference."""
        style_ids, frame_ids = kwargs['style_ids'], kwargs['frame_ids']
        z = self(style_ids=style_ids, frame_ids=frame_ids, type=kwargs['data_type'])
        sid = style_ids.to(z.device).long()
        mu, logvar = self.style_latents_mu.detach()[sid], self.style_latents_logvar.detach()[sid]
        return torch.sum((z - mu) ** 2 / (torch.exp(0.5 * logvar) + 1e-3), -1).mean()

    differentiable = False

    def trainable(self, on=True):
        """Training side: the gather becomes differentiable w.r.t. the latent table and mu (tgtc_latents_backward)."""
        self.differentiable = bool(on)
        return self
