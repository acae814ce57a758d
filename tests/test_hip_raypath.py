"""GPU parity of the per-ray HIP kernels (through the C ABI) against the reference goldens and the oracle.

Tolerances are written next to each check.  float64 ray generation is expected bit-exact (same
operation order, no FMA contraction); float32 kernels to ~1 ulp; the inverse-CDF sampler is
ill-conditioned where the pdf is ~1e-5 (empty space), see the comment there.
"""
import numpy as np
import pytest
import torch

from oracle import raymarch, rays
from tgtc_style_amd import synth

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def maxdiff(a, b):
    return float((torch.as_tensor(a).double().cpu() - torch.as_tensor(np.asarray(b)).double()).abs().max())


def test_gen_rays_golden(golden):
    from tgtc_style_amd import utils
    g = golden("g1_rays")
    for tag, (H, W) in {"a": (12, 16), "b": (16, 16)}.items():
        focal = synth.fern_intrinsics(H, W)
        for pi in (0, 37):
            for pa in (0, 1):
                key = "%s_p%d_%d" % (tag, pi, pa)
                o, d = utils.gen_rays(H, W, focal, synth.spiral_pose(pi), pixel_alignment=bool(pa), ndc=False)
                assert maxdiff(o, g[key + "_o"].reshape(-1, 3)) == 0.0
                assert maxdiff(d, g[key + "_d"].reshape(-1, 3)) <= 1e-15      # float64, same op order
                o, d = utils.gen_rays(H, W, focal, synth.spiral_pose(pi), pixel_alignment=bool(pa), ndc=True)
                assert maxdiff(o, g[key + "_ndc_o"].reshape(-1, 3)) <= 1e-14
                assert maxdiff(d, g[key + "_ndc_d"].reshape(-1, 3)) <= 1e-14


def test_gen_rays_shard_equals_whole():
    """A pixel sub-range (what one rank generates) is bit-identical to the same rows of the whole frame."""
    from tgtc_style_amd import utils
    H, W = 24, 40
    focal = synth.fern_intrinsics(H, W)
    o, d = utils.gen_rays(H, W, focal, synth.spiral_pose(5))
    o2, d2 = utils.gen_rays(H, W, focal, synth.spiral_pose(5), first_pixel=311, n=401)
    assert torch.equal(o[311:712], o2) and torch.equal(d[311:712], d2)
    ro, rd = rays.frame_rays_ndc(H, W, focal, synth.spiral_pose(5))
    assert maxdiff(o, ro) <= 1e-14 and maxdiff(d, rd) <= 1e-14


def test_sample_coarse_golden(golden):
    from tgtc_style_amd import utils
    g = golden("g2_coarse")
    ro, rd = dev(g["rays_o"]), dev(g["rays_d"])
    for n in (64, 128):
        pts, ts = utils.sampling_pts_uniform(ro, rd, N_samples=n, near=0., far=1., perturb=False)
        assert pts.dtype == torch.float64 and ts.dtype == torch.float32
        assert maxdiff(ts, g["ts_%d" % n]) == 0.0                  # linspace reproduced bit-exactly
        assert maxdiff(pts, g["pts_%d" % n]) <= 1e-15
        pts, ts = utils.sampling_pts_uniform(ro, rd, N_samples=n, near=0., far=1., perturb=True, jitter=dev(g["jit_%d" % n]))
        assert maxdiff(ts, g["ts_jit_%d" % n]) == 0.0
        assert maxdiff(pts, g["pts_jit_%d" % n]) <= 1e-15
    # perturb=True without an explicit jitter stays stratified and ascending
    _, ts = utils.sampling_pts_uniform(ro, rd, N_samples=64, near=0., far=1., perturb=True)
    assert bool((ts[:, 1:] >= ts[:, :-1]).all()) and float(ts.min()) >= 0 and float(ts.max()) <= 1


def test_posenc_golden(golden):
    from tgtc_style_amd import models
    g = golden("g3_embed")
    x = dev(g["x"])
    e10, e4 = models.Embedder(3, 9, 10), models.Embedder(3, 3, 4)
    # float64 sin/cos then float32 cast, like the reference: half an ulp of float32
    assert maxdiff(e10(x), g["pe10_f64"].astype(np.float32)) <= 6e-8
    assert maxdiff(e4(x), g["pe4_f64"].astype(np.float32)) <= 6e-8
    # float32 input: the argument x*512 carries float32 rounding, sinf is ~1-2 ulp
    assert maxdiff(e10(x.float()), g["pe10_f32in"]) <= 5e-7


def test_composite_golden(golden):
    from tgtc_style_amd import utils
    g = golden("g5_composite")
    for s in ("", "2"):
        r, t, w = utils.alpha_composition(dev(g["rgb" + s]), dev(g["sigma" + s]), dev(g["ts" + s]), 0)
        # fp32 scan in a different association order than cumprod: a few ulp on values <= 1
        assert maxdiff(w, g["weights" + s]) <= 5e-7
        assert maxdiff(r, g["rgb_exp" + s]) <= 1e-6
        assert maxdiff(t, g["t_exp" + s]) <= 1e-6
    w = utils.alpha_composition(dev(g["rgb"]), dev(g["sigma"]), dev(g["ts"]), 0)[2].cpu().numpy()
    assert np.all(w[1] == 0) and np.all(w[2] == 0)


@pytest.mark.parametrize("N", [2, 3, 63, 64, 65, 128, 192, 255, 256])
def test_composite_ragged_lengths(N):
    """Every per-lane run length C and ragged tails, against the oracle.  (N = 1 is degenerate in the
    reference: utils.py:367-369 builds the 1e10 tail from an empty slice, so all outputs are empty.)"""
    from tgtc_style_amd import utils
    rng = np.random.default_rng(N)
    R = 9
    rgb = rng.uniform(0, 1, (R, N, 3)).astype(np.float32)
    sigma = (rng.standard_normal((R, N)) * 30).astype(np.float32)
    ts = np.sort(rng.uniform(0, 1, (R, N)).astype(np.float32), -1)
    r0, t0, w0 = raymarch.composite(torch.from_numpy(rgb), torch.from_numpy(sigma), torch.from_numpy(ts))
    r, t, w = utils.alpha_composition(dev(rgb), dev(sigma), dev(ts), 0)
    assert maxdiff(w, w0) <= 5e-7 and maxdiff(r, r0) <= 1e-6 and maxdiff(t, t0) <= 1e-6


def test_composite_empty():
    from tgtc_style_amd import utils
    r, t, w = utils.alpha_composition(torch.empty(0, 8, 3).cuda(), torch.empty(0, 8).cuda(), torch.empty(0, 8).cuda(), 0)
    assert r.shape == (0, 3) and t.shape == (0,) and w.shape == (0, 8)


def test_sample_fine_golden(golden):
    from tgtc_style_amd import utils
    g = golden("g6_fine")
    for N in (64, 128):
        tag = "_%d" % N
        ro, rd, ts, w = (dev(g[k + tag]) for k in ("rays_o", "rays_d", "ts", "w"))
        pts, tv = utils.sampling_pts_fine_torch(ro, rd, ts, w, 64)
        assert tv.shape == (ro.shape[0], N + 64)
        assert bool((tv[:, 1:] >= tv[:, :-1]).all())
        # cdf: float64 running sum like ATen's CPU cumsum, but the float32 normaliser (sum of weights) is
        # reduced in a different order -> pdf differs by ~1 ulp; a sample at cdf position u moves by
        # d(cdf)/pdf, and pdf is as small as 1e-5/sum in empty bins: |dt| <~ 1e-7/1e-5 * bin(1/64) ~ 2e-4 worst
        # case, typically < 1e-6.
        d = (tv.double().cpu() - torch.from_numpy(g["tvals" + tag]).double()).abs()
        assert float(d.max()) <= 2e-4, float(d.max())
        assert float(d.median()) <= 1e-7
        p = ro[:, None, :] + rd[:, None, :] * tv[..., None].double()
        assert maxdiff(pts, p.cpu()) <= 1e-15


def test_sample_fine_is_a_permutation_of_coarse_plus_new():
    """Size-independent property: the output contains every coarse depth exactly once (rank sort is a
    permutation), at full size (128 coarse + 64 fine, many rays)."""
    from tgtc_style_amd import utils
    rng = np.random.default_rng(0)
    R, N, NF = 4096, 128, 64
    ro = dev(rng.uniform(-1, 1, (R, 3)))
    rd = dev(rng.uniform(-1, 1, (R, 3)))
    _, ts = utils.sampling_pts_uniform(ro, rd, N_samples=N, near=0., far=1., perturb=True)
    w = dev((rng.uniform(0, 1, (R, N)) ** 6).astype(np.float32))
    _, tv = utils.sampling_pts_fine_torch(ro, rd, ts, w, NF)
    assert bool((tv[:, 1:] >= tv[:, :-1]).all())
    # multiset check via sorted concatenation against the oracle's sampler
    _, tv0 = raymarch.sample_fine(ro.cpu(), rd.cpu(), ts.cpu(), w.cpu(), NF)
    # The reference algorithm is discontinuous where a bin's cdf step is < 1e-5 (utils.py:604-605 sets
    # denom = 1, pinning the sample to the bin's lower edge): a 1-ulp difference in the cdf can move such a
    # sample by one whole bin.  So: all but a handful agree to 1e-5, none differs by more than one bin.
    d = (tv.cpu() - tv0).abs()
    assert float((d > 1e-5).float().mean()) <= 1e-3, float((d > 1e-5).float().mean())
    assert float(d.max()) <= 1.0 / (N - 1) + 1e-4
    # every coarse depth survives bit-exactly
    merged = torch.cat([tv, ts], 1).sort(1)[0]
    dup = (merged[:, 1:] == merged[:, :-1]).sum(1)
    assert int(dup.min()) >= N


def test_latents_golden(golden):
    from tgtc_style_amd import models
    g = golden("g7_style")
    lat = models.StyleLatents_variational(style_num=2, frame_num=20, latent_dim=32)
    lat.load_state_dict({k: torch.from_numpy(v) for k, v in synth.latents_state(4, style_num=2, frame_num=20).items()})
    lat = lat.cuda()
    sid, fid = torch.from_numpy(g["style_ids"]), torch.from_numpy(g["frame_ids"])
    for sc in (0.0, 1.0, 0.35):
        lat.sigma_scale = sc
        out = lat(style_ids=sid, frame_ids=fid, type="llff")
        assert maxdiff(out, g["latents_s%g" % sc]) <= 1e-7     # one fused multiply-add of difference at most
    with pytest.raises(IndexError):
        lat(style_ids=torch.tensor([1]), frame_ids=torch.tensor([7 * 40]), type="llff")


def test_image_epilogue_golden_and_edges(golden):
    """a13 image epilogue (rendering.py:66-71 / :202-207 / :358-361): uint8 images bit-identical to what the reference
    wrote (g10), to the oracle on full-size frames, and on the edge cases of the numpy casts."""
    from oracle import image
    from tgtc_style_amd import utils
    g = golden("g10_image")
    frames = int(g["frames"])
    rgb8, depth8 = utils.frames_to_uint8(dev(g["rgb"]), dev(g["t"]), frames)
    assert np.array_equal(rgb8.cpu().numpy().reshape(g["rgb8"].shape), g["rgb8"])
    assert np.array_equal(depth8.cpu().numpy().reshape(g["depth8"].shape), g["depth8"])
    # BASELINE frame size, several frames, both eps conventions
    rng = np.random.default_rng(21)
    rgb = rng.uniform(0, 1, (3 * 160000, 3)).astype(np.float32)
    t = rng.uniform(0.1, 0.9, 3 * 160000).astype(np.float32)
    for eps in (1e-7, 0.0):
        a, b = utils.frames_to_uint8(dev(rgb), dev(t), 3, eps)
        ra, rb = image.frames_to_uint8(rgb, t, 3, eps)
        assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(b.cpu().numpy(), rb)
    # edges: colours a hair above 1 wrap to 0 (256 -> uint8), exact 1.0 -> 255, a constant depth plane -> 0 / eps = 0
    # (and 0/0 = NaN -> 0 without eps), a single pixel per frame, either output alone
    rgb = np.array([[1.0, 1.004, 0.0], [0.999999, 0.5, 1.0039216]], np.float32)
    t = np.array([0.25, 0.25], np.float32)
    for eps in (1e-7, 0.0):
        for frames in (1, 2):
            a, b = utils.frames_to_uint8(dev(rgb), dev(t), frames, eps)
            ra, rb = image.frames_to_uint8(rgb, t, frames, eps)
            assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(b.cpu().numpy(), rb), (eps, frames)
    assert ra[0, 0].tolist() == [255, 0, 0]
    only_rgb, none = utils.frames_to_uint8(dev(rgb), None, 1)
    assert none is None and np.array_equal(only_rgb.cpu().numpy(), image.frames_to_uint8(rgb, t, 1)[0])
    none, only_t = utils.frames_to_uint8(None, dev(np.array([0.5, 0.1, 0.9], np.float32)), 1)
    assert none is None and only_t.cpu().numpy().tolist() == [[127, 0, 254]]
