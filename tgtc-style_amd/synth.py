"""Seeded synthetic weights, poses and frames (no dataset or checkpoint is available offline).

Framework independent (numpy PCG64), so golden generation, the oracle, the HIP path and
bench.py all see bit-identical parameters.  State-dict key names and shapes are the
reference's (models.py:63-94, :120-163, :475-486; transformer.py:13-44;
tctrans.py:13-99) so the dicts load straight into the reference modules.

Seeds follow SURVEY.md section 8d: 0 coarse NeRF, 1 fine NeRF, 2 concat MLP, 3 style MLP, 4 latents,
5 ViT (Xavier-uniform on matrices, transformer.py:41-44), 6 patch-embed / CNN decoder / VGG.
"""
import math

import numpy as np

PE_COOR = 63   # 3 + 3*2*10  (models.py:190-191)
PE_DIR = 27    # 3 + 3*2*4   (models.py:192-193)
LATENT = 32    # config.py:86
HIDDEN_GAIN = math.sqrt(6.0)   # He-uniform: nn.Linear's default bound 1/sqrt(fan_in) x sqrt(6) keeps ReLU activations O(1)
SIGMA_GAIN = 500.0             # sigma head scaled up so densities are peaky and the fine sampler is exercised
SIGMA_SHIFT = 20.0             # ... centred so ~30% of space is occupied and the first hit varies per ray
PE_DECAY = 0.35                # input weights of band k scaled by 2^(-PE_DECAY*k): a 1/f^0.35-ish spectrum like a trained
                               # NeRF's (spectral bias).  With PE_DECAY = 0 the density is white noise along a ray and the
                               # coarse->fine chain amplifies 1e-6 perturbations to ~1e-3 (even fp32 vs fp64 of the reference).


def _damp_pe(w, col0, decay, n_freqs=10):
    """Scale the columns that multiply band k of a 63-wide point encoding starting at column col0."""
    if decay:
        for k in range(n_freqs):
            w[:, col0 + 3 + 6 * k: col0 + 9 + 6 * k] *= np.float32(2.0 ** (-decay * k))
    return w


def _linear(rng, out_f, in_f, gain=1.0):
    b = 1.0 / math.sqrt(in_f)
    w = rng.uniform(-b, b, size=(out_f, in_f)).astype(np.float32) * np.float32(gain)
    bias = rng.uniform(-b, b, size=(out_f,)).astype(np.float32) * np.float32(gain)
    return w, bias


def nerf_state(seed, depth=8, width=256, skips=(4,), use_viewdir=True, sigma_gain=SIGMA_GAIN,
               sigma_shift=SIGMA_SHIFT, gain=HIDDEN_GAIN, pe_decay=PE_DECAY):
    """StyleNerf state dict (keys `net.*`).  Shapes: models.py:76-93."""
    rng = np.random.default_rng(seed)
    sd, dim = {}, PE_COOR
    for i in range(depth):
        sd["net.base_layers.%d.weight" % i], sd["net.base_layers.%d.bias" % i] = _linear(rng, width, dim, gain)
        dim = width
        if i in skips and i != depth - 1:
            dim += PE_COOR
    w, b = _linear(rng, 1, dim, sigma_gain)
    # zero-mean sigma weights decouple the density level from the (positive) mean activation, so
    # sigma ~ N(sigma_shift, (0.2*sigma_gain)^2) for every seed
    sd["net.sigma_layer.weight"] = (w - w.mean()).astype(np.float32)
    sd["net.sigma_layer.bias"] = (b * np.float32(0) + np.float32(sigma_shift)).astype(np.float32)
    sd["net.base_remap_layer.weight"], sd["net.base_remap_layer.bias"] = _linear(rng, 256, dim, gain)
    d = 256 + PE_DIR if use_viewdir else 256
    sd["net.rgb_layers.0.weight"], sd["net.rgb_layers.0.bias"] = _linear(rng, width // 2, d, gain)
    sd["net.rgb_layers.1.weight"], sd["net.rgb_layers.1.bias"] = _linear(rng, 3, width // 2, gain)
    _damp_pe(sd["net.base_layers.0.weight"], 0, pe_decay)
    for i in skips:
        if i + 1 < depth:
            _damp_pe(sd["net.base_layers.%d.weight" % (i + 1)], 0, pe_decay)   # cat(pe, h): pe first (models.py:98-99)
    return sd


def heavy_tailed(sd, seed, family):
    """A NeRF state dict with heavy-tailed numbers (what trained MLPs look like; the seeded nets above are uniform).

    'rows' (log-normal, sigma = 2 octaves) and 'outliers' (four features x64, two x1/64 per layer) use the
    positive-scaling symmetry of a ReLU layer -- row i of layer l and its bias times s_i, column i of every consumer of
    that feature times 1/s_i, s_i a power of two -- so the network computes the SAME function (the scene stays as well
    conditioned as the base one) while per-feature activation ranges and per-column weight ranges spread by up to 2^12.
    'elements' multiplies every trunk weight by its own log-normal factor: a different function (callers re-centre the
    density head, see tests/probes/emu_mx_e2e.py recalibrate_sigma)."""
    if family == "base":
        return sd
    rng = np.random.default_rng(1000 + seed)
    sd = {k: v.copy() for k, v in sd.items()}
    if family == "elements":
        for i in range(1, 8):
            w = sd["net.base_layers.%d.weight" % i]
            f = np.exp(rng.normal(0.0, 0.6, w.shape)).astype(np.float32)
            sd["net.base_layers.%d.weight" % i] = w * f / np.float32(math.exp(0.18))   # E[f] = e^0.18: keep the activation scale
        return sd
    if family not in ("rows", "outliers"):
        raise ValueError(family)
    for i in range(8):
        n = 256
        if family == "rows":
            s = np.exp2(np.rint(rng.normal(0.0, 2.0, n))).astype(np.float32)
        else:
            s = np.ones(n, np.float32)
            idx = rng.choice(n, 6, replace=False)
            s[idx[:4]], s[idx[4:]] = 64.0, 1.0 / 64.0
        name = "net.base_layers.%d" % i
        sd[name + ".weight"] = sd[name + ".weight"] * s[:, None]
        sd[name + ".bias"] = sd[name + ".bias"] * s
        cons = [("net.base_layers.%d" % (i + 1), 63 if i == 4 else 0)] if i < 7 else [("net.sigma_layer", 0), ("net.base_remap_layer", 0)]
        for c, c0 in cons:       # layer 5 reads cat(pe(63), h) (models.py:98-99)
            sd[c + ".weight"][:, c0:c0 + n] *= (1.0 / s)[None, :]
    return sd


def heavy_tailed_style(concat_sd, style_sd, seed, family):
    """The stylised chain's two MLPs (StyleMLP_before_concat, StyleMLP_Wild_multilayers) with heavy-tailed numbers: the 'rows' /
    'outliers' families of `heavy_tailed` applied through the same ReLU scaling symmetry -- row i of a hidden layer and its
    bias times s_i (a power of two), column i of its consumer times 1/s_i -- so both nets compute the SAME function.  Input
    layouts (models.py:137-147, :165-180): every layer reads cat(h, latent [, x]) with h in columns 0..255; the style MLP's
    layer 0 reads cat(base_remap, concat_features, x, latent), the concat MLP's output in columns 256..511."""
    if family == "base":
        return concat_sd, style_sd
    if family not in ("rows", "outliers"):
        raise ValueError(family)
    rng = np.random.default_rng(3000 + seed)
    c = {k: v.copy() for k, v in concat_sd.items()}
    st = {k: v.copy() for k, v in style_sd.items()}

    def scales(n=256):
        if family == "rows":
            return np.exp2(np.rint(rng.normal(0.0, 2.0, n))).astype(np.float32)
        s = np.ones(n, np.float32)
        idx = rng.choice(n, 6, replace=False)
        s[idx[:4]], s[idx[4:]] = 64.0, 1.0 / 64.0
        return s

    def apply(prod, name, cons, cname, c0):
        s = scales()
        prod[name + ".weight"] = prod[name + ".weight"] * s[:, None]
        prod[name + ".bias"] = prod[name + ".bias"] * s
        cons[cname + ".weight"][:, c0:c0 + 256] *= (1.0 / s)[None, :]
    for i in range(4):
        apply(c, "layers.%d" % i, c, "layers.%d" % (i + 1), 0)
    apply(c, "layers.4", st, "layers.0", 256)
    for i in range(7):
        apply(st, "layers.%d" % i, st, "layers.%d" % (i + 1), 0)
    return c, st


def nerf_state_adversarial(seed):
    """White-noise density (no spectral decay, mostly empty space): the ill-conditioned stress scene."""
    return nerf_state(seed, sigma_shift=-40.0, pe_decay=0.0)


def concat_state(seed=2, style_D=8, width=256, skip=4, gain=HIDDEN_GAIN, pe_decay=PE_DECAY):
    """StyleMLP_before_concat state dict (keys `layers.*`).  models.py:121-135: the loop breaks
    after appending the skip layer, so there are skip+1 layers."""
    rng = np.random.default_rng(seed)
    sd, dim = {}, PE_COOR + LATENT
    for i in range(style_D - 1):
        if i == skip:
            dim += PE_COOR
        sd["layers.%d.weight" % i], sd["layers.%d.bias" % i] = _linear(rng, width, dim, gain)
        if i == 0:
            _damp_pe(sd["layers.0.weight"], 0, pe_decay)                      # cat(x, latent)
        if i == skip:
            _damp_pe(sd["layers.%d.weight" % i], width + LATENT, pe_decay)    # cat(h, latent, x)
            break
        dim = width + LATENT
    return sd


def style_state(seed=3, style_D=8, width=256, skip=4, gain=HIDDEN_GAIN, pe_decay=PE_DECAY):
    """StyleMLP_Wild_multilayers state dict.  models.py:150-163."""
    rng = np.random.default_rng(seed)
    sd, dim = {}, PE_COOR + 512 + LATENT
    for i in range(style_D - 1):
        if i == skip:
            dim += PE_COOR
        sd["layers.%d.weight" % i], sd["layers.%d.bias" % i] = _linear(rng, width, dim, gain)
        if i == 0:
            _damp_pe(sd["layers.0.weight"], 512, pe_decay)                    # cat(concated(512), x, latent)
        if i == skip:
            _damp_pe(sd["layers.%d.weight" % i], width + LATENT, pe_decay)    # cat(h, latent, x)
        dim = width + LATENT
    sd["layers.%d.weight" % (style_D - 1)], sd["layers.%d.bias" % (style_D - 1)] = _linear(rng, 3, width + LATENT, gain)
    return sd


def latents_state(seed=4, style_num=1, frame_num=20, dim=LATENT):
    """StyleLatents_variational parameters ~ N(0,1).  models.py:482-486."""
    rng = np.random.default_rng(seed)
    return {"latents": rng.standard_normal((style_num, frame_num, dim)).astype(np.float32),
            "style_latents_mu": rng.standard_normal((style_num, dim)).astype(np.float32),
            "style_latents_logvar": rng.standard_normal((style_num, dim)).astype(np.float32)}


def vae_state(seed=9, data_dim=1024, latent_dim=LATENT, W=512, D=4):
    """VAE parameters under the reference's key names (models.py:371-457: encoder.fc_layers.i, encoder.fc_layer_mu,
    encoder.fc_layer_log_var, decoder.fc_layers.i, decoder.output_layer), nn.Linear default init scale."""
    rng = np.random.default_rng(seed)
    sd = {}
    dim = data_dim
    for i in range(D - 1):
        sd["encoder.fc_layers.%d.weight" % i], sd["encoder.fc_layers.%d.bias" % i] = _linear(rng, W, dim, math.sqrt(2.0))
        dim = W
    sd["encoder.fc_layer_mu.weight"], sd["encoder.fc_layer_mu.bias"] = _linear(rng, latent_dim, dim)
    sd["encoder.fc_layer_log_var.weight"], sd["encoder.fc_layer_log_var.bias"] = _linear(rng, latent_dim, dim)
    dim = latent_dim
    for i in range(D - 1):
        sd["decoder.fc_layers.%d.weight" % i], sd["decoder.fc_layers.%d.bias" % i] = _linear(rng, W, dim, math.sqrt(2.0))
        dim = W
    sd["decoder.output_layer.weight"], sd["decoder.output_layer.bias"] = _linear(rng, data_dim, dim)
    return sd


def _xavier(rng, *shape):
    fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-a, a, size=shape).astype(np.float32)


def transformer_state(seed=5, d=512, ff=2048, n_enc=3, n_dec=3):
    """Transformer state dict, 142 keys.  Matrices Xavier-uniform
    (transformer.py:41-44); biases small uniform (not zero, so bias handling is exercised);
    LayerNorm weight ~ 1 +- 0.1."""
    rng = np.random.default_rng(seed)
    sd = {}

    def small(n):
        return rng.uniform(-0.05, 0.05, size=(n,)).astype(np.float32)

    def norm(p):
        sd[p + ".weight"] = (1.0 + rng.uniform(-0.1, 0.1, size=(d,))).astype(np.float32)
        sd[p + ".bias"] = small(d)

    def attn(p):
        sd[p + "in_proj_weight"] = _xavier(rng, 3 * d, d)
        sd[p + "in_proj_bias"] = small(3 * d)
        sd[p + "out_proj.weight"] = _xavier(rng, d, d)
        sd[p + "out_proj.bias"] = small(d)

    def ffn(p):
        sd[p + "linear1.weight"] = _xavier(rng, ff, d)
        sd[p + "linear1.bias"] = small(ff)
        sd[p + "linear2.weight"] = _xavier(rng, d, ff)
        sd[p + "linear2.bias"] = small(d)

    for enc in ("encoder_c", "encoder_s"):
        for i in range(n_enc):
            p = "%s.layers.%d." % (enc, i)
            sd[p + "qk.weight"] = _xavier(rng, 2 * d, d)
            sd[p + "qkv.weight"] = _xavier(rng, 3 * d, d)
            attn(p + "self_attn.")
            ffn(p)
            norm(p + "norm1")
            norm(p + "norm2")
    for i in range(n_dec):
        p = "decoder.layers.%d." % i
        attn(p + "self_attn.")
        attn(p + "multihead_attn.")
        ffn(p)
        norm(p + "norm1")
        norm(p + "norm2")
        norm(p + "norm3")
    norm("decoder.norm")
    sd["new_ps.weight"] = _xavier(rng, d, d, 1, 1)
    sd["new_ps.bias"] = small(d)
    return sd


def _conv(rng, out_c, in_c, k):
    b = 1.0 / math.sqrt(in_c * k * k)
    return (rng.uniform(-b, b, size=(out_c, in_c, k, k)).astype(np.float32),
            rng.uniform(-b, b, size=(out_c,)).astype(np.float32))


# Sequential index -> (out_ch, in_ch) for the 3x3 convs of the CNN decoder (tctrans.py:36-66)
DECODER_SHAPES = {1: (256, 512), 5: (256, 256), 8: (256, 256), 11: (256, 256), 14: (128, 256),
                  18: (128, 128), 21: (64, 128), 25: (64, 64), 28: (3, 64)}
# Sequential index -> (out_ch, in_ch, k) for vgg[:31] (tctrans.py:68-99)
VGG_SHAPES = {0: (3, 3, 1), 2: (64, 3, 3), 5: (64, 64, 3), 9: (128, 64, 3), 12: (128, 128, 3),
              16: (256, 128, 3), 19: (256, 256, 3), 22: (256, 256, 3), 25: (256, 256, 3), 29: (512, 256, 3)}


def embed_state(seed=6):
    rng = np.random.default_rng(seed)
    w, b = _conv(rng, 512, 3, 8)
    return {"proj.weight": w, "proj.bias": b}


def decoder_state(seed=7, gain=1.6):
    """CNN decoder convs; default-init weights shrink activations ~x0.58 per ReLU conv, so a gain
    keeps the 9-conv chain O(1) (otherwise outputs collapse to the biases and test nothing)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for idx, (o, i) in DECODER_SHAPES.items():
        w, b = _conv(rng, o, i, 3)
        sd["%d.weight" % idx], sd["%d.bias" % idx] = w * np.float32(gain), b
    return sd


def vgg_state(seed=8, gain=1.6):
    rng = np.random.default_rng(seed)
    sd = {}
    for idx, (o, i, k) in VGG_SHAPES.items():
        w, b = _conv(rng, o, i, k)
        sd["%d.weight" % idx], sd["%d.bias" % idx] = w * np.float32(gain if k == 3 else 1.0), b
    return sd


def spiral_pose(i, n=120, radius=0.25):
    """A small seeded camera motion about identity: 3x4 float32 c2w, looking down -z, in the spirit of
    load_llff.render_path_spiral (load_llff.py:145-154) but closed-form and data-free."""
    a = 2.0 * math.pi * i / n
    eye = np.array([radius * math.cos(a), -radius * math.sin(a), -0.3 * radius * math.sin(0.5 * a)])
    target = np.array([0.0, 0.0, -4.0])
    z = eye - target
    z /= np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.concatenate([np.stack([x, y, z], 1), eye[:, None]], 1).astype(np.float32)


def fern_intrinsics(H, W):
    """fern-like focal: focal = 0.82*W."""
    return 0.82 * W


def style_image(seed, H, W):
    return np.random.default_rng(seed).uniform(0, 1, size=(1, 3, H, W)).astype(np.float32)
