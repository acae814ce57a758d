"""GPU: the single persistent ray kernel behind tgtc_render_rays_plain (render_fused.hip) against
  * the reference's own renders (golden g8: cal_geometry chain on 64 rays, 128c+64f and 64c+64f, with and without
    stratified jitter), at the north-star tolerance 1e-3;
  * the chain of per-sample kernels (tgtc_render_rays_plain_chain) on the same rays: same network arithmetic, the
    compositing scan associates differently, so agreement is to float32 rounding;
  * itself under re-sharding: any sub-range of rays, any ray count (not a multiple of the 8 rays of a workgroup
    step), reproduces the same bits -- rays are independent and a ray's arithmetic does not depend on the wave,
    workgroup or launch that renders it.
"""
import numpy as np
import pytest
import torch

from tgtc_style_amd import synth

pytestmark = pytest.mark.gpu


def T(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


class Args:
    use_viewdir, act_type = True, "relu"
    embed_freq_coor, embed_freq_dir = 10, 4
    netdepth = netdepth_fine = 8
    netwidth = netwidth_fine = 256
    precision = "fp16x3"


def renderer(prec_c, prec_f, fused):
    from tgtc_style_amd import models, rendering
    nets = []
    for (seed, mode), prec in zip(((0, "coarse"), (1, "fine")), (prec_c, prec_f)):
        m = models.StyleNerf(type("A", (Args,), {"precision": prec}), mode=mode)
        m.load_state_dict(T(synth.nerf_state(seed)))
        nets.append(m.cuda())
    # (True here means THE single kernel: RayRenderer's own True lets the library pick, and for fp16x3 + fp16mx it picks the split path)
    return rendering.RayRenderer(nets[0], nets[1], fused="single" if fused is True else fused)


PAIRS = [("fp16x3", "fp16x3"), ("fp16x3", "fp16mx"), ("fp16", "fp16")]
LIMIT = {"fp16x3": 1e-3, "fp16mx": 1e-3, "fp16": 2e-2}      # against the reference (fp16 is the documented fast mode)
# fused vs chain: the two composite scans associate differently, so the coarse weights differ in the last bit and the
# fine depths move by ~1e-7.  fp16x3 follows such a perturbation smoothly (4e-7 end to end); the fp6 roundings of fp16mx
# and the fp16 roundings of the fast mode turn it into their own rounding noise.
VS_CHAIN = {"fp16x3": 1e-5, "fp16mx": 1e-4, "fp16": 2e-3}


@pytest.mark.parametrize("prec_c,prec_f", PAIRS)
@pytest.mark.parametrize("nc,nf", [(128, 64), (64, 64)])
def test_fused_render_golden(golden, prec_c, prec_f, nc, nf):
    g = golden("g8_end_to_end")
    tag = "_%dc%df" % (nc, nf)
    ro, rd = torch.from_numpy(g["rays_o" + tag]).cuda(), torch.from_numpy(g["rays_d" + tag]).cuda()
    fused, chain = renderer(prec_c, prec_f, True), renderer(prec_c, prec_f, False)
    assert fused._fused_shape(nc, nf)
    for jt, jit in (("", None), ("_jit", torch.from_numpy(g["jit" + tag]).cuda())):
        a = fused.render(ro, rd, nc, nf, near=0., far=1., jitter=jit)
        b = chain.render(ro, rd, nc, nf, near=0., far=1., jitter=jit)
        e_chain = max(float((a["rgb"] - b["rgb"]).abs().max()), float((a["t"] - b["t"]).abs().max()))
        e_ref = 0.0
        if jit is None:      # the reference's plain chain (cal_geometry) never jitters; the jittered case is pinned by the chain
            e_ref = max(float((a["rgb"].cpu() - torch.from_numpy(g["plain_rgb" + tag])).abs().max()),
                        float((a["t"].cpu() - torch.from_numpy(g["plain_t" + tag])).abs().max()))
        print(prec_c, prec_f, tag, jt, "vs reference %.2e  vs chain %.2e" % (e_ref, e_chain))
        assert e_ref <= LIMIT[prec_f] and e_chain <= VS_CHAIN[prec_f]


def test_default_path_of_the_headline_pair_is_the_split_path(golden):
    """RayRenderer(fused=True) hands the workspace over for coarse fp16x3 + fine fp16mx and the library renders through the
    per-sample kernels (fine pass: the two-tile kernel, mlp_nerf_mx2.hip): the same bits as the forced chain; 'single' still
    reaches the ray kernel and agrees to the fused-vs-chain tolerance."""
    from tgtc_style_amd import rendering
    g = golden("g8_end_to_end")
    ro, rd = torch.from_numpy(g["rays_o_128c64f"]).cuda(), torch.from_numpy(g["rays_d_128c64f"]).cuda()
    single, chain = renderer("fp16x3", "fp16mx", True), renderer("fp16x3", "fp16mx", False)
    auto = rendering.RayRenderer(chain.coarse, chain.fine)
    assert auto._split_is_faster() and auto.fused is True
    a, b, c = auto.render(ro, rd, 128, 64), chain.render(ro, rd, 128, 64), single.render(ro, rd, 128, 64)
    assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["t"], b["t"])
    assert float((a["rgb"] - c["rgb"]).abs().max()) <= VS_CHAIN["fp16mx"]
    with pytest.raises(ValueError):
        rendering.RayRenderer(chain.coarse, chain.fine, fused="single").render(ro, rd, 128, 64, want_coarse=True)


@pytest.mark.parametrize("prec_c,prec_f", PAIRS)
def test_fused_render_ray_counts_and_shards(prec_c, prec_f):
    from tgtc_style_amd import utils
    H = W = 400
    ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(3), first_pixel=33 * W, n=2077)
    r = renderer(prec_c, prec_f, True)
    whole = r.render(ro, rd, 128, 64)
    assert bool(torch.isfinite(whole["rgb"]).all()) and bool(torch.isfinite(whole["t"]).all())
    assert float(whole["rgb"].min()) >= 0 and float(whole["rgb"].max()) <= 1 + 1e-5
    for lo, hi in ((0, 1), (5, 18), (1000, 2077), (7, 2056)):      # 1, 13, 1077, 2049 rays
        part = r.render(ro[lo:hi].contiguous(), rd[lo:hi].contiguous(), 128, 64)
        assert torch.equal(part["rgb"], whole["rgb"][lo:hi]) and torch.equal(part["t"], whole["t"][lo:hi]), (lo, hi)
    chain = renderer(prec_c, prec_f, False).render(ro, rd, 128, 64)
    e = max(float((chain["rgb"] - whole["rgb"]).abs().max()), float((chain["t"] - whole["t"]).abs().max()))
    print(prec_c, prec_f, "2077 rays, fused vs chain %.2e" % e)
    assert e <= VS_CHAIN[prec_f]


def test_fused_falls_back_to_the_chain():
    """Sample counts the ray kernel does not tile (n_coarse not a multiple of 16) and requests for the coarse image run
    the per-sample chain behind the same entry point."""
    from tgtc_style_amd import utils
    from tgtc_style_amd import rendering
    single = renderer("fp16x3", "fp16x3", True)
    r = rendering.RayRenderer(single.coarse, single.fine)
    with pytest.raises(ValueError):
        single.render(*utils.gen_rays(4, 4, synth.fern_intrinsics(4, 4), synth.spiral_pose(1)), 100, 28)
    assert not r._fused_shape(100, 28) and r._fused_shape(128, 64) and not r._fused_shape(208, 48)
    ro, rd = utils.gen_rays(16, 16, synth.fern_intrinsics(16, 16), synth.spiral_pose(1))
    a = r.render(ro, rd, 100, 28)
    b = renderer("fp16x3", "fp16x3", False).render(ro, rd, 100, 28)
    assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["t"], b["t"])
    c = r.render(ro, rd, 128, 64, want_coarse=True)
    assert "rgb_coarse" in c and bool(torch.isfinite(c["rgb_coarse"]).all())


# ---------------------------------------------------------------------------------------------------------------------
# The stylised ray kernel behind tgtc_render_rays_styled (render_styled_fused.hip)
def styled_renderer(fused):
    from tgtc_style_amd import models, rendering
    nets = []
    for seed, mode in ((0, "coarse"), (1, "fine")):
        m = models.StyleNerf(Args, mode=mode)
        m.load_state_dict(T(synth.nerf_state(seed)))
        nets.append(m.cuda())
    cm, sm = models.StyleMLP_before_concat(type("A", (Args,), {"style_D": 8, "vae_latent": 32})), \
        models.StyleMLP_Wild_multilayers(type("A", (Args,), {"style_D": 8, "vae_latent": 32}))
    cm.load_state_dict(T(synth.concat_state(2))), sm.load_state_dict(T(synth.style_state(3)))
    return rendering.RayRenderer(nets[0], nets[1], style=models.StylePair(cm.cuda(), sm.cuda()), fused=fused)


@pytest.mark.parametrize("nc,nf", [(128, 64), (64, 64), (16, 16)])
def test_fused_styled_render_against_the_chain(nc, nf):
    """One launch of the stylised ray kernel against the chain of per-sample kernels (tgtc_render_rays_styled_chain: same
    network arithmetic, compositing scan associated differently), with and without jitter, ragged ray counts; the chain
    itself is pinned to the reference's own stylised renders (tests/test_hip_style.py, g8)."""
    from tgtc_style_amd import utils
    H = W = 400
    R = 1003
    ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(2), first_pixel=201 * W + 17, n=R)
    gen = torch.Generator(device="cuda").manual_seed(5)
    z = torch.randn(R, 32, device="cuda", generator=gen)
    fused, chain = styled_renderer(True), styled_renderer(False)
    assert fused._fused_styled_shape(nc, nf)
    for jit in (None, torch.rand(R, nc, device="cuda", generator=gen)):
        a = fused.render(ro, rd, nc, nf, near=0., far=1., jitter=jit, z=z)
        b = chain.render(ro, rd, nc, nf, near=0., far=1., jitter=jit, z=z)
        assert bool(torch.isfinite(a["rgb"]).all()) and bool(torch.isfinite(a["t"]).all())
        e = max(float((a["rgb"] - b["rgb"]).abs().max()), float((a["t"] - b["t"]).abs().max()))
        print("styled %dc+%df jitter %s: fused vs chain %.2e" % (nc, nf, jit is not None, e))
        assert e <= VS_CHAIN["fp16x3"]
    # re-sharding: a ray's bits do not depend on the launch that renders it
    whole = fused.render(ro, rd, nc, nf, z=z)
    for lo, hi in ((0, 1), (5, 18), (500, 1003)):
        part = fused.render(ro[lo:hi].contiguous(), rd[lo:hi].contiguous(), nc, nf, z=z[lo:hi].contiguous())
        assert torch.equal(part["rgb"], whole["rgb"][lo:hi]) and torch.equal(part["t"], whole["t"][lo:hi]), (lo, hi)
