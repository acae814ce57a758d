"""Oracle: per-ray sampling and compositing (PyTorch CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference utils.py:509-531 (sampling_pts_uniform), utils.py:573-609
(sampling_pts_fine_torch / sample_pdf) and utils.py:354-386 (alpha_composition).
"""
import torch


def sample_coarse(rays_o, rays_d, n_samples, near, far, jitter=None, dtype=torch.float32):
    """Stratified coarse samples.  utils.py:509-531.

    ts = linspace(0,1,N)*(far-near)+near in float32 (:512-514), broadcast to all rays.
    With jitter (the reference draws U(0,1) via nn.init.uniform_, :519-520; here it is an
    explicit [R,N] tensor so results are reproducible): each sample moves inside the
    interval bounded by the midpoints to its neighbours (:521-524).
    pts = o + ts*d inherits the dtype of the rays (float64 in the render path).
    Returns pts [R,N,3], ts [R,N] (float32).
    """
    R = rays_o.shape[0]
    ts = torch.linspace(0, 1, n_samples, dtype=dtype).unsqueeze(0).expand(R, n_samples)
    ts = ts * (far - near) + near
    if jitter is not None:
        mid = (ts[..., 1:] + ts[..., :-1]) / 2
        hi = torch.cat([mid, ts[..., -1:]], -1)
        lo = torch.cat([ts[..., :1], mid], -1)
        ts = lo + (hi - lo) * jitter
    pts = rays_o[:, None, :] + ts[..., None] * rays_d[:, None, :]
    return pts, ts


def composite(rgb, sigma, ts):
    """Alpha compositing at sigma_noise_std = 0.  utils.py:354-386.

    delta_i = t_{i+1}-t_i, last delta = 1e10 (:367-369); NOT scaled by |d|.
    alpha = 1-exp(-relu(relu(sigma))*delta) (:365,:376).
    T_i = prod_{j<i}(1-alpha_j+1e-10) via cumprod of [1, 1-alpha+1e-10][:-1] (:378).
    Returns rgb_exp [R,3], t_exp [R], weights [R,N].
    """
    delta = ts[..., 1:] - ts[..., :-1]
    delta = torch.cat([delta, torch.full_like(delta[..., :1], 1e10)], -1)
    dens = torch.relu(torch.relu(sigma))
    alpha = 1. - torch.exp(-dens * delta)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    return (w[..., None] * rgb).sum(-2), (w * ts).sum(-1), w


def inverse_cdf(bins, weights, n_fine):
    """Deterministic inverse-CDF sampling.  utils.py:583-609 with det=True.

    weights + 1e-5 -> pdf -> cdf with a leading 0 (:584-587); u = linspace(0,1,n_fine)
    (:589-591); searchsorted(right=True) (:595); below = max(0, i-1), above = min(last, i)
    (:596-597); gather cdf and bins at both (:601-602); denom < 1e-5 -> 1 (:604-605);
    linear interpolation (:606-607).
    bins [R,B], weights [R,B-1] -> samples [R,n_fine]
    """
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0., 1., n_fine, dtype=cdf.dtype).expand(list(cdf.shape[:-1]) + [n_fine]).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = (idx - 1).clamp(min=0)
    hi = idx.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def sample_fine(rays_o, rays_d, ts, weights, n_fine):
    """Hierarchical fine samples.  utils.py:573-580.

    bins = midpoints of ts (:574); pdf from weights[:,1:-1] (:575); the n_fine new depths are
    merged with the coarse ones by a sort (:577); pts = o + d*t (:578).
    Returns pts [R,N+n_fine,3], t_vals [R,N+n_fine] ascending.
    """
    mid = 0.5 * (ts[..., 1:] + ts[..., :-1])
    new_t = inverse_cdf(mid, weights[..., 1:-1], n_fine)
    t_all = torch.sort(torch.cat([ts, new_t], -1), -1)[0]
    pts = rays_o[:, None, :] + rays_d[:, None, :] * t_all[..., None]
    return pts, t_all
