"""GPU: WHOLE frames at the BASELINE sizes (128 coarse + 64 fine samples per ray) through the fused chains
(tgtc_render_rays_plain / tgtc_render_rays_styled), checked against the CPU oracle on rays spread over the whole
frame -- the first rows, the middle and the LAST rows, where index arithmetic at ~3.7e7 samples per launch would go
wrong first -- and, for BASELINE config 4 (trex 504x378 over 8 ranks), that each rank's pixel range reproduces the
bits of the whole-frame render.

    config 2: fern  400x400 = 160 000 rays, plain          config 3: the same frame, stylised chain
    config 4: trex  504x378 = 190 512 rays, 8 contiguous ray ranges of 23 814 (parallel.shard_range)
    config 5: every LLFF-shaped frame in configs/*.txt renders through the same two entry points
"""
import numpy as np
import pytest
import torch

from oracle import fields
from tgtc_style_amd import parallel, synth

pytestmark = pytest.mark.gpu

NC, NF = 128, 64
SHAPES = {"fern": (400, 400), "trex": (378, 504)}       # (H, W)


def T(sd, dtype=None):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype or torch.float32) for k, v in sd.items()}


class Args:
    use_viewdir, act_type = True, "relu"
    embed_freq_coor, embed_freq_dir = 10, 4
    netdepth = netdepth_fine = 8
    netwidth = netwidth_fine = 256
    style_D, vae_latent = 8, 32
    precision = "fp16x3"


def nets(precision="fp16x3"):
    from tgtc_style_amd import models
    a = type("A", (Args,), {"precision": precision})
    out = []
    for seed, mode in ((0, "coarse"), (1, "fine")):
        m = models.StyleNerf(a, mode=mode)
        m.load_state_dict(T(synth.nerf_state(seed)))
        out.append(m.cuda())
    return a, out


def check_against_oracle(label, rgb, t, render, ro, rd, coarse, median_bound=1e-5, mixed=True):
    """Every spot-check ray within 1e-3 of an admissible output of the reference (tests/conditioning.py: the float32 oracle,
    or the branch the reference itself takes when its arithmetic is perturbed at the rounding level); no ray is exempt.  The
    median against the float32 oracle must sit at the precision mode's own level (fp32 rounding for fp16x3, the fp6
    correction's ~1e-4 for a fine pass in fp16mx)."""
    import conditioning
    # `coarse`: the coarse StyleNerf, for the stage certificate of rays the fixed variant set does not explain
    conditioning.check(label, rgb, t, render, ro, rd, tol=1e-3, median_bound=median_bound, mixed=mixed, n_fine=NF,
                       stages=lambda sel: conditioning.hip_stages(coarse, ro[sel].cuda(), rd[sel].cuda(), NC, NF))


def spot_indices(H, W, per_band=400):
    """>= 1 200 ray indices: the first two rows, two rows around the middle, and the LAST two rows of the frame."""
    n = H * W
    bands = [(0, 2 * W), (n // 2 - W, n // 2 + W), (n - 2 * W, n)]
    idx = np.concatenate([np.linspace(lo, hi - 1, per_band).astype(np.int64) for lo, hi in bands] + [[n - 1]])
    return torch.from_numpy(np.unique(idx))


def nets_mixed():
    """bench.py's default precision: coarse pass fp16x3, fine pass fp16 + two block-scaled fp6 corrections."""
    (_, (coarse, _)), (_, (_, fine)) = nets("fp16x3"), nets("fp16mx")
    return coarse, fine


@pytest.mark.parametrize("precision", ["fp16x3", "fp16x3+fp16mx"])
@pytest.mark.parametrize("scene", ["fern", "trex"])
def test_whole_frame_plain(scene, precision):
    from tgtc_style_amd import rendering, utils
    H, W = SHAPES[scene]
    n = H * W
    coarse, fine = nets()[1] if precision == "fp16x3" else nets_mixed()
    r = rendering.RayRenderer(coarse, fine)
    ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(9))
    assert ro.shape == (n, 3)
    out = r.render(ro, rd, NC, NF)
    rgb, t = out["rgb"], out["t"]
    assert rgb.shape == (n, 3) and bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(t).all())
    idx = spot_indices(H, W)
    assert idx.numel() >= 1200 and int(idx[-1]) == n - 1
    check_against_oracle("%s %dx%d plain %s" % (scene, W, H, precision), rgb[idx], t[idx],
                         lambda o, d, dc, df, sel, **kw: fields.render_plain(T(synth.nerf_state(0), dc), T(synth.nerf_state(1), df), o, d, NC, NF,
                                                                        dtype=dc, dtype_fine=df, **kw),
                         ro[idx].cpu(), rd[idx].cpu(), coarse, median_bound=1e-5 if precision == "fp16x3" else 2e-4)
    if scene != "trex":
        return
    # config 4: eight contiguous ray ranges; every rank's range alone reproduces the whole-frame bits
    assert n == 190512 and parallel.shard_range(n, 7, 8) == (166698, 190512)
    for rank in range(8):
        lo, hi = parallel.shard_range(n, rank, 8)
        assert hi - lo == 23814
        so, sd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(9), first_pixel=lo, n=hi - lo)
        assert torch.equal(so, ro[lo:hi]) and torch.equal(sd, rd[lo:hi])
        part = r.render(so, sd, NC, NF)
        assert torch.equal(part["rgb"], rgb[lo:hi]) and torch.equal(part["t"], t[lo:hi]), "rank %d" % rank


@pytest.mark.parametrize("scene", ["fern", "trex"])
def test_whole_frame_styled(scene):
    from tgtc_style_amd import models, rendering, utils
    H, W = SHAPES[scene]
    n = H * W
    a, (coarse, fine) = nets()
    cm, sm = models.StyleMLP_before_concat(a), models.StyleMLP_Wild_multilayers(a)
    cm.load_state_dict(T(synth.concat_state(2)))
    sm.load_state_dict(T(synth.style_state(3)))
    lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    lat.load_state_dict(T(synth.latents_state(4)))
    lat = lat.cuda()
    lat.sigma_scale = 1.0
    r = rendering.RayRenderer(coarse, fine, models.StylePair(cm.cuda(), sm.cuda()))
    ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(17))
    z = lat(style_ids=torch.zeros(n, dtype=torch.long), frame_ids=torch.full((n,), 17, dtype=torch.long), type="llff")
    out = r.render(ro, rd, NC, NF, z=z)
    rgb, t = out["rgb"], out["t"]
    assert rgb.shape == (n, 3) and bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(t).all())
    idx = spot_indices(H, W, per_band=200)      # the stylised oracle costs 2.5x the plain one
    raw = [synth.nerf_state(0), synth.nerf_state(1), synth.concat_state(2), synth.style_state(3)]
    zi = z[idx].cpu()

    def oracle(o, d, dc, df, sel, **kw):    # the stylised chain has one dtype (dc); sel = subset of the spot-check rays
        zz = (zi if sel is None else zi[sel]).to(dc)
        w = [T(sd, dc) for sd in raw]
        return fields.render_styled(w[0], w[1], w[2], w[3], o, d, zz, NC, NF, dtype=dc, **kw)
    check_against_oracle("%s %dx%d styled" % (scene, W, H), rgb[idx], t[idx], oracle, ro[idx].cpu(), rd[idx].cpu(), coarse, mixed=False)
    if scene != "trex":
        return
    for rank in (0, 3, 7):          # first, an interior and the last rank of the 8-way split
        lo, hi = parallel.shard_range(n, rank, 8)
        part = r.render(ro[lo:hi].contiguous(), rd[lo:hi].contiguous(), NC, NF, z=z[lo:hi].contiguous())
        assert torch.equal(part["rgb"], rgb[lo:hi]) and torch.equal(part["t"], t[lo:hi]), "rank %d" % rank
