"""Oracle: positional encoding, NeRF MLP, the two style MLPs and the latent table (PyTorch CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Networks are evaluated functionally from state dicts that use the reference's key names, so
the same dict can be loaded into the reference modules (golden generation) and packed for the
HIP library (product).

Follows reference models.py:24-60 (Embedder), :63-117 (MLP_style), :182-223 (StyleNerf),
:120-147 (StyleMLP_before_concat), :149-180 (StyleMLP_Wild_multilayers),
:475-506 (StyleLatents_variational) and rendering.py:18-56 / :109-178 (the two render chains).
"""
import torch
import torch.nn.functional as F

from . import raymarch


def posenc(x, n_freqs):
    """[x, sin(x*2^0), cos(x*2^0), ..., sin(x*2^(L-1)), cos(x*2^(L-1))].  models.py:46-60.

    Frequency bands are 2**linspace(0, L-1, L) (:40), each sin/cos block is as wide as x.
    Computed in the dtype of x (float64 in the render path); callers cast to float32
    afterwards (models.py:219-220).
    """
    parts = [x]
    for k in range(n_freqs):
        f = float(2.0 ** k)
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def nerf_mlp(sd, pts_enc, dirs_enc, depth=8, skips=(4,), use_viewdir=True, prefix="net."):
    """MLP_style.forward for ReLU nets (sigma_mul = 0).  models.py:95-117.

    trunk: h = relu(L0 pe); for i in 0..depth-2: (i in skips -> h = cat(pe, h)); h = relu(L_{i+1} h)
    sigma = Lsigma h (:103); base_remap = relu(Lremap h) (:106);
    rgb = sigmoid(Lc1 relu(Lc0 cat(base_remap, dirs))) (:107-111).
    """
    h = torch.relu(_lin(sd, prefix + "base_layers.0", pts_enc))
    for i in range(depth - 1):
        if i in skips:
            h = torch.cat([pts_enc, h], -1)
        h = torch.relu(_lin(sd, prefix + "base_layers.%d" % (i + 1), h))
    sigma = _lin(sd, prefix + "sigma_layer", h).squeeze(-1)
    remap = torch.relu(_lin(sd, prefix + "base_remap_layer", h))
    feat = torch.cat([remap, dirs_enc], -1) if use_viewdir else remap
    feat = torch.relu(_lin(sd, prefix + "rgb_layers.0", feat))
    rgb = torch.sigmoid(_lin(sd, prefix + "rgb_layers.1", feat))
    return {"rgb": rgb, "base_remap": remap, "pts": pts_enc, "sigma": sigma}


def style_nerf(sd, pts, dirs, freq_coor=10, freq_dir=4, dtype=torch.float32, **kw):
    """StyleNerf.forward: encode (in the input dtype), cast to float32, run the MLP.  models.py:216-223.
    dtype=float64 (with float64 weights in sd) evaluates the same formulas in double precision: the conditioning probe
    of the whole-frame tests, not the reference's arithmetic."""
    pe = posenc(pts, freq_coor).to(dtype)
    de = posenc(dirs, freq_dir).to(dtype)
    out = nerf_mlp(sd, pe, de, **kw)
    out["dirs"] = de
    return out


def concat_mlp(sd, x, latent, skip=4):
    """StyleMLP_before_concat.forward.  models.py:137-147.

    The constructor stops adding layers at the skip (models.py:129-132) so there are skip+1 layers;
    every layer sees cat(h, latent) and layer `skip` additionally cat(.., x).
    """
    h = x
    i = 0
    while ("layers.%d.weight" % i) in sd:
        h = torch.cat([h, latent], -1)
        if i == skip:
            h = torch.cat([h, x], -1)
        h = torch.relu(_lin(sd, "layers.%d" % i, h))
        i += 1
    return {"concat_features": h}


def style_mlp(sd, x, concated, latent, skip=4):
    """StyleMLP_Wild_multilayers.forward.  models.py:165-180."""
    n = 0
    while ("layers.%d.weight" % n) in sd:
        n += 1
    h = torch.cat([concated, x], -1)
    for i in range(n - 1):
        h = torch.cat([h, latent], -1)
        if i == skip:
            h = torch.cat([h, x], -1)
        h = torch.relu(_lin(sd, "layers.%d" % i, h))
    h = torch.cat([h, latent], -1)
    return {"rgb": torch.sigmoid(_lin(sd, "layers.%d" % (n - 1), h))}


def latents_forward(sd, style_ids, frame_ids, sigma_scale=1.0, llff=True):
    """StyleLatents_variational.forward.  models.py:490-506.

    flat = style*frame_num + frame (:492); for llff the flattened table is tiled 7x before the
    gather (:496) so that render-pose ids beyond frame_num wrap; out = mu + sigma_scale*(lat-mu) (:504-505).
    """
    lat = sd["latents"]
    frame_num, dim = lat.shape[1], lat.shape[2]
    flat = style_ids * frame_num + frame_ids
    table = lat.reshape(-1, dim)
    if llff:
        table = table.repeat(7, 1)
    z = table[flat]
    mu = sd["style_latents_mu"][style_ids]
    return mu + sigma_scale * (z - mu)


def vae_encode(sd, x, depth=4):
    """VAE.encode(x, various=False) (reference models.py:390-395, :450-453): D-1 Linear+ReLU layers, then the mu and
    log-variance heads.  x [S,1024] float32 -> (mu [S,32], log_var [S,32])."""
    h = x
    for i in range(depth - 1):
        h = torch.relu(_lin(sd, "encoder.fc_layers.%d" % i, h))
    return _lin(sd, "encoder.fc_layer_mu", h), _lin(sd, "encoder.fc_layer_log_var", h)


def render_plain(sd_coarse, sd_fine, rays_o, rays_d, n_coarse, n_fine, near=0., far=1., dtype=torch.float32, dtype_fine=None,
                 ts_fine=None):
    """The cal_geometry chain, one chunk.  rendering.py:27-51 (perturb=False, det fine sampling).

    Returns dict with coarse and fine rgb_exp / t_exp / weights plus the fine t values.
    dtype / dtype_fine = float64 (with float64 state dicts): the coarse / fine half of the chain in double precision -- the
    conditioning probes of the whole-frame tests (tests/conditioning.py), not the reference's arithmetic.
    ts_fine [R, n_coarse + n_fine] (optional): evaluate the fine network and its compositing at THESE merged depths instead
    of the ones sampled here (conditional parity of tests/conditioning.py: the sampler's branch taken as given).
    """
    dtype_fine = dtype_fine or dtype
    pts, ts = raymarch.sample_coarse(rays_o, rays_d, n_coarse, near, far, dtype=dtype)
    dirs = rays_d[:, None, :].expand(-1, n_coarse, -1)
    c = style_nerf(sd_coarse, pts, dirs, dtype=dtype)
    rgb_c, t_c, w_c = raymarch.composite(c["rgb"], c["sigma"], ts)
    pts_f, ts_f = raymarch.sample_fine(rays_o, rays_d, ts.to(dtype_fine), w_c.to(dtype_fine), n_fine)
    if ts_fine is not None:
        ts_f = ts_fine.to(dtype_fine)
        pts_f = rays_o[:, None, :] + rays_d[:, None, :] * ts_f[..., None]        # raymarch.sample_fine's last line (utils.py:578)
    dirs = rays_d[:, None, :].expand(-1, n_coarse + n_fine, -1)
    f = style_nerf(sd_fine, pts_f, dirs, dtype=dtype_fine)
    rgb_f, t_f, w_f = raymarch.composite(f["rgb"], f["sigma"], ts_f)
    return {"rgb_coarse": rgb_c, "t_coarse": t_c, "w_coarse": w_c, "ts_fine": ts_f,
            "sigma_fine": f["sigma"], "rgb_fine": rgb_f, "t_fine": t_f, "w_fine": w_f}


def _styled_pass(sd_nerf, sd_concat, sd_style, pts, dirs, z, dtype=torch.float32):
    """One pass of the stylised chain.  rendering.py:122-142 (= :158-175 for the fine pass).

    z [R,32] is the per-ray latent; the concat MLP receives it as is, the style MLP receives its
    mean over the 32 channels broadcast back to 32 (rendering.py:126,139).
    """
    n = pts.shape[1]
    out = style_nerf(sd_nerf, pts, dirs, dtype=dtype)
    z1 = z[:, None, :].expand(-1, n, -1)
    cf = concat_mlp(sd_concat, out["pts"], z1)["concat_features"]
    both = torch.cat([out["base_remap"], cf], -1)
    zbar = z.mean(1, keepdim=True)[:, :, None].expand(-1, n, z.shape[-1])
    rgb = style_mlp(sd_style, out["pts"], both, zbar)["rgb"]
    return rgb, out["sigma"]


def render_styled(sd_coarse, sd_fine, sd_concat, sd_style, rays_o, rays_d, z,
                  n_coarse, n_fine, near=0., far=1., jitter=None, dtype=torch.float32, ts_fine=None):
    """The render_style chain, one batch.  rendering.py:118-178.  dtype=float64, ts_fine: see render_plain."""
    pts, ts = raymarch.sample_coarse(rays_o, rays_d, n_coarse, near, far, jitter, dtype=dtype)
    dirs = rays_d[:, None, :].expand(-1, n_coarse, -1)
    rgb, sig = _styled_pass(sd_coarse, sd_concat, sd_style, pts, dirs, z, dtype)
    rgb_c, t_c, w_c = raymarch.composite(rgb, sig, ts)
    pts_f, ts_f = raymarch.sample_fine(rays_o, rays_d, ts, w_c, n_fine)
    if ts_fine is not None:
        ts_f = ts_fine.to(ts.dtype)
        pts_f = rays_o[:, None, :] + rays_d[:, None, :] * ts_f[..., None]
    dirs = rays_d[:, None, :].expand(-1, n_coarse + n_fine, -1)
    rgb, sig = _styled_pass(sd_fine, sd_concat, sd_style, pts_f, dirs, z, dtype)
    rgb_f, t_f, w_f = raymarch.composite(rgb, sig, ts_f)
    return {"rgb_coarse": rgb_c, "t_coarse": t_c, "w_coarse": w_c, "ts_fine": ts_f,
            "rgb_fine": rgb_f, "t_fine": t_f, "w_fine": w_f}
