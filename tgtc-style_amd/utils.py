"""Per-ray operators with the reference's `utils.py` names and call signatures, backed by the HIP
library (reference utils.py:354-386, :435-456, :509-531, :573-609).

Inputs are CUDA/HIP tensors; outputs are freshly allocated tensors, as in the reference.  There is
no CPU implementation here: without a GPU or without libtgtc_hip.so these functions raise.
"""
import torch

from . import hip


def _f64(t):
    return t.to(torch.float64).contiguous()


def _f32(t):
    return t.to(torch.float32).contiguous()


def sampling_pts_uniform(rays_o, rays_d, N_samples=64, near=0., far=1.05, harmony=False, perturb=False, jitter=None):
    """reference utils.py:509-531.  Returns (pts [R,N,3] float64, ts [R,N] float32).

    `perturb=True` draws the stratified jitter with torch.rand on the device (the reference uses
    nn.init.uniform_, utils.py:519-520); pass `jitter` ([R,N] in [0,1)) to make it reproducible.
    """
    if harmony:
        raise NotImplementedError("harmony (inverse-depth) sampling is unused by the reference render paths")
    hip.require_gpu(rays_o, rays_d)
    lib = hip.load()
    rays_o, rays_d = _f64(rays_o), _f64(rays_d)
    R = rays_o.shape[0]
    if perturb and jitter is None:
        jitter = torch.rand(R, N_samples, device=rays_o.device, dtype=torch.float32)
    if jitter is not None:
        jitter = _f32(jitter)
        assert jitter.shape == (R, N_samples)
    pts = torch.empty(R, N_samples, 3, device=rays_o.device, dtype=torch.float64)
    ts = torch.empty(R, N_samples, device=rays_o.device, dtype=torch.float32)
    hip.check(lib.tgtc_sample_coarse(hip.ptr(rays_o), hip.ptr(rays_d), R, N_samples, float(near), float(far),
                                     hip.ptr(jitter), hip.ptr(pts), hip.ptr(ts), hip.stream()))
    return pts, ts


def sampling_pts_fine_torch(rays_o, rays_d, ts, weights, N_samples_fine=64):
    """reference utils.py:573-580 (deterministic inverse-CDF + sorted merge).
    Returns (pts [R,N+Nf,3] float64, t_vals [R,N+Nf] float32 ascending)."""
    hip.require_gpu(rays_o, rays_d, ts, weights)
    lib = hip.load()
    rays_o, rays_d, ts, weights = _f64(rays_o), _f64(rays_d), _f32(ts), _f32(weights)
    R, N = ts.shape
    T = N + N_samples_fine
    pts = torch.empty(R, T, 3, device=ts.device, dtype=torch.float64)
    tv = torch.empty(R, T, device=ts.device, dtype=torch.float32)
    hip.check(lib.tgtc_sample_fine(hip.ptr(rays_o), hip.ptr(rays_d), hip.ptr(ts), hip.ptr(weights), R, N,
                                   N_samples_fine, hip.ptr(pts), hip.ptr(tv), hip.stream()))
    return pts, tv


class _Composite(torch.autograd.Function):
    """alpha_composition as a differentiable op on the HIP kernels: forward tgtc_composite_train, backward
    tgtc_composite_backward (gradients w.r.t. rgb and sigma; the depths and the injected noise carry none, as in the
    reference, where they come from the sampler and from torch.randn)."""

    @staticmethod
    def forward(ctx, rgb, sigma, ts, noise, white_bkgd):
        lib = hip.load()
        R, N = sigma.shape
        rgb_exp = torch.empty(R, 3, device=rgb.device, dtype=torch.float32)
        t_exp = torch.empty(R, device=rgb.device, dtype=torch.float32)
        w = torch.empty(R, N, device=rgb.device, dtype=torch.float32)
        hip.check(lib.tgtc_composite_train(hip.ptr(rgb), hip.ptr(sigma), hip.ptr(ts), hip.ptr(noise), int(white_bkgd), R, N,
                                           hip.ptr(rgb_exp), hip.ptr(t_exp), hip.ptr(w), hip.stream()))
        ctx.save_for_backward(rgb, sigma, ts, noise)
        ctx.white_bkgd = int(white_bkgd)
        return rgb_exp, t_exp, w

    @staticmethod
    def backward(ctx, g_rgb, g_t, g_w):
        lib = hip.load()
        rgb, sigma, ts, noise = ctx.saved_tensors
        R, N = sigma.shape
        d_rgb = torch.empty_like(rgb) if ctx.needs_input_grad[0] else None
        d_sigma = torch.empty_like(sigma) if ctx.needs_input_grad[1] else None
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        hip.check(lib.tgtc_composite_backward(hip.ptr(rgb), hip.ptr(sigma), hip.ptr(ts), hip.ptr(noise), ctx.white_bkgd, R, N,
                                              hip.ptr(c(g_rgb)), hip.ptr(c(g_t)), hip.ptr(c(g_w)), hip.ptr(d_rgb),
                                              hip.ptr(d_sigma), hip.stream()))
        return d_rgb, d_sigma, None, None, None


def alpha_composition(pts_rgb, pts_sigma, t_values, sigma_noise_std=0., white_bkgd=False, noise=None):
    """reference utils.py:354-386.  Returns (rgb_exp [R,3], t_exp [R], weights [R,N]).  `noise` ([R,N], optional) is the
    density regulariser already drawn (the reference draws randn * sigma_noise_std inside, :371-374); differentiable
    w.r.t. pts_rgb and pts_sigma when they require grad (the reference's training loops, train_tgtcs.py:218-309)."""
    hip.require_gpu(pts_rgb, pts_sigma, t_values)
    rgb, sigma, ts = _f32(pts_rgb), _f32(pts_sigma), _f32(t_values)
    if noise is None and sigma_noise_std > 0:
        noise = torch.randn(sigma.shape, device=sigma.device) * sigma_noise_std
    if noise is not None:
        noise = _f32(noise)
    if noise is None and not white_bkgd and not (rgb.requires_grad or sigma.requires_grad):
        lib = hip.load()
        R, N = sigma.shape
        rgb_exp = torch.empty(R, 3, device=rgb.device, dtype=torch.float32)
        t_exp = torch.empty(R, device=rgb.device, dtype=torch.float32)
        w = torch.empty(R, N, device=rgb.device, dtype=torch.float32)
        hip.check(lib.tgtc_composite(hip.ptr(rgb), hip.ptr(sigma), hip.ptr(ts), R, N, hip.ptr(rgb_exp), hip.ptr(t_exp),
                                     hip.ptr(w), hip.stream()))
        return rgb_exp, t_exp, w
    return _Composite.apply(rgb, sigma, ts, noise, bool(white_bkgd))


def batchify(fn, chunk=1024 * 32):
    """reference utils.py:435-456: split every kwarg on dim 0 into `chunk`-row pieces and concatenate
    each entry of the returned dicts."""
    if chunk is None:
        return fn

    def chunked(**kwargs):
        first = next(iter(kwargs.values()))
        pieces = {}
        for lo in range(0, first.shape[0], chunk):
            ret = fn(**{k: v[lo:lo + chunk] for k, v in kwargs.items()})
            for k, v in ret.items():
                pieces.setdefault(k, []).append(v)
        return {k: torch.cat(v, 0) for k, v in pieces.items()}

    return chunked


def gen_rays(H, W, focal, c2w, first_pixel=0, n=None, pixel_alignment=False, ndc=True, ndc_near=1.0, device="cuda"):
    """Rays of pixels [first_pixel, first_pixel+n) of an H x W frame on the device: reference
    dataset.py:33-42 + :44-61 with K = [[f,0,W/2],[0,f,H/2],[0,0,1]] (dataset.py:392-398).
    Returns (rays_o, rays_d) float64 [n,3]."""
    hip.require_gpu()
    lib = hip.load()
    import numpy as np
    c2w = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32)[:3, :4])
    n = H * W - first_pixel if n is None else n
    o = torch.empty(n, 3, device=device, dtype=torch.float64)
    d = torch.empty(n, 3, device=device, dtype=torch.float64)
    hip.check(lib.tgtc_gen_rays(H, W, float(focal), float(focal), 0.5 * W, 0.5 * H, c2w.ctypes.data,
                                int(pixel_alignment), int(ndc), float(ndc_near), first_pixel, n, hip.ptr(o),
                                hip.ptr(d), hip.stream()))
    return o, d


def frames_to_uint8(rgb, t, frames, eps=1e-7):
    """Image epilogue of the render drivers on the device (rendering.py:66-71, :202-207, :358-361; utils.py:463 to8b):
    per-frame min-max normalised depth and colour as uint8(int32(x * 255)).  rgb [frames*P,3], t [frames*P] float32
    CUDA tensors -> (uint8 [frames,P,3], uint8 [frames,P]) CUDA tensors; either input may be None."""
    lib = hip.load()
    ref = rgb if rgb is not None else t
    pixels = ref.shape[0] // frames
    assert ref.is_cuda and ref.shape[0] == frames * pixels
    rgb8 = depth8 = None
    if rgb is not None:
        rgb = rgb.detach().contiguous().float()
        rgb8 = torch.empty(frames, pixels, 3, dtype=torch.uint8, device=rgb.device)
    if t is not None:
        t = t.detach().contiguous().float()
        depth8 = torch.empty(frames, pixels, dtype=torch.uint8, device=t.device)
    hip.check(lib.tgtc_image_epilogue(hip.ptr(rgb) if rgb is not None else None, hip.ptr(t) if t is not None else None,
                                      frames, pixels, float(eps), hip.ptr(rgb8) if rgb8 is not None else None,
                                      hip.ptr(depth8) if depth8 is not None else None, hip.stream()))
    return rgb8, depth8
