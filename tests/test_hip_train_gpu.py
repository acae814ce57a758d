"""GPU: training-side forms of alpha compositing (reference utils.py:371-376, 381-384 and
the autograd path of train_tgtcs.py:218-309): the density-noise regulariser and the white background in the forward
kernel, and the backward kernel against torch.autograd on the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import raymarch

pytestmark = pytest.mark.gpu


def _inputs(R, N, seed):
    rng = np.random.default_rng(seed)
    rgb = torch.from_numpy(rng.random((R, N, 3)).astype(np.float32))
    sigma = torch.from_numpy((rng.standard_normal((R, N)) * 8).astype(np.float32))
    ts = torch.from_numpy(np.sort(rng.random((R, N)).astype(np.float32), -1))
    noise = torch.from_numpy(rng.standard_normal((R, N)).astype(np.float32))
    return rgb, sigma, ts, noise


@pytest.mark.parametrize("N", [7, 64, 128, 192])
def test_composite_noise_and_white_background(N):
    from tgtc_style_amd import utils
    rgb, sigma, ts, noise = _inputs(33, N, N)
    for nz, white in ((noise, False), (None, True), (noise, True)):
        got = utils.alpha_composition(rgb.cuda(), sigma.cuda(), ts.cuda(), white_bkgd=white, noise=None if nz is None else nz.cuda())
        ref_rgb, ref_t, ref_w = raymarch.composite(rgb, sigma if nz is None else sigma + nz, ts)
        if white:
            ref_rgb = ref_rgb + (1. - ref_w.sum(-1, keepdim=True))          # utils.py:381-384
        assert float((got[0].cpu() - ref_rgb).abs().max()) <= 2e-6
        assert float((got[1].cpu() - ref_t).abs().max()) <= 2e-6 and float((got[2].cpu() - ref_w).abs().max()) <= 2e-6


@pytest.mark.parametrize("N,white", [(5, False), (64, False), (128, True), (192, False)])
def test_composite_backward_matches_autograd_on_the_oracle(N, white):
    from tgtc_style_amd import utils
    rgb, sigma, ts, noise = _inputs(21, N, 100 + N)
    g = np.random.default_rng(7)
    g_rgb = torch.from_numpy(g.standard_normal((21, 3)).astype(np.float32))
    g_t = torch.from_numpy(g.standard_normal(21).astype(np.float32))
    g_w = torch.from_numpy(g.standard_normal((21, N)).astype(np.float32))
    # reference gradients: float64 autograd through the oracle's alpha_composition
    r64, s64 = rgb.double().requires_grad_(True), sigma.double().requires_grad_(True)
    o_rgb, o_t, o_w = raymarch.composite(r64, s64 + noise.double(), ts.double())
    if white:
        o_rgb = o_rgb + (1. - o_w.sum(-1, keepdim=True))
    loss = (o_rgb * g_rgb.double()).sum() + (o_t * g_t.double()).sum() + (o_w * g_w.double()).sum()
    ref_dr, ref_ds = torch.autograd.grad(loss, (r64, s64))
    # HIP: the same loss through the autograd.Function
    dr, ds = rgb.cuda().requires_grad_(True), sigma.cuda().requires_grad_(True)
    h_rgb, h_t, h_w = utils.alpha_composition(dr, ds, ts.cuda(), white_bkgd=white, noise=noise.cuda())
    ((h_rgb * g_rgb.cuda()).sum() + (h_t * g_t.cuda()).sum() + (h_w * g_w.cuda()).sum()).backward()
    e_r = float((dr.grad.cpu().double() - ref_dr).abs().max() / ref_dr.abs().max())
    e_s = float((ds.grad.cpu().double() - ref_ds).abs().max() / ref_ds.abs().max())
    print("N=%d white=%s: d rgb %.2e  d sigma %.2e (max-norm relative)" % (N, white, e_r, e_s))
    assert e_r <= 1e-5 and e_s <= 1e-4
    # only sigma needs a gradient (the stylised training step freezes the NeRF colour head's inputs): rgb grad is skipped
    ds2 = sigma.cuda().requires_grad_(True)
    out = utils.alpha_composition(rgb.cuda(), ds2, ts.cuda(), noise=noise.cuda())
    out[2].sum().backward()
    assert ds2.grad is not None and bool(torch.isfinite(ds2.grad).all())
