"""GPU parity of the fused PE + NeRF MLP kernels and the plain render chain, through the C ABI.

Tolerance (BASELINE.json north_star): RGB / sigma within 1e-3 relative of the fp32 reference.
"relative" is taken against the largest magnitude of the reference tensor (max-norm relative error):
    err(a, ref) = max|a - ref| / max|ref|
which is the meaningful reading for sigma (pre-activation values cross zero, where a pointwise
ratio is unbounded).  Mode fp16x3 (the default, parity mode) must meet 1e-3 everywhere and in fact
sits near 1e-5; mode fp16mx (fp16 product + two block-scaled fp6 correction products) must meet 1e-3 as
well and sits near 1e-4; mode fp16 (single fp16 MFMA product) is the documented fast mode and is held to 1e-2.
"""
import numpy as np
import pytest
import torch

from oracle import fields, raymarch
from tgtc_style_amd import synth

pytestmark = pytest.mark.gpu

TOL = {"fp16x3": 1e-3, "fp16mx": 1e-3, "fp16": 1e-2}
# what each path actually achieves; guards against silent precision regressions
TIGHT = {"fp16x3": 5e-5, "fp16mx": 5e-4, "fp16": 1e-2}


def T(sd, cuda=False):
    out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    return {k: v.cuda() for k, v in out.items()} if cuda else out


def rel(a, ref):
    a, ref = torch.as_tensor(a).double().cpu(), torch.as_tensor(np.asarray(ref)).double()
    return float((a - ref).abs().max() / ref.abs().max())


class Args:
    use_viewdir = True
    act_type = "relu"
    embed_freq_coor, embed_freq_dir = 10, 4
    netdepth = netdepth_fine = 8
    netwidth = netwidth_fine = 256
    style_D, vae_latent = 8, 32
    precision = "fp16x3"


def make_nerf(seed, mode, precision):
    from tgtc_style_amd import models
    a = type("A", (Args,), {"precision": precision})
    m = models.StyleNerf(a, mode=mode)
    m.load_state_dict(T(synth.nerf_state(seed)))
    return m.cuda()


@pytest.mark.parametrize("precision", ["fp16x3", "fp16mx", "fp16"])
def test_stylenerf_forward_golden(golden, precision):
    """StyleNerf.forward (models.py:216-223) on the reference's own outputs (g4)."""
    g = golden("g4_nerf")
    ro, rd, ts = (torch.from_numpy(g[k]).cuda() for k in ("rays_o", "rays_d", "ts"))
    pts = ro[:, None, :] + ts[..., None].double() * rd[:, None, :]
    dirs = rd[:, None, :].expand(-1, ts.shape[1], -1)
    for name, seed in (("coarse", 0), ("fine", 1)):
        m = make_nerf(seed, name, precision)
        out = m(pts=pts, dirs=dirs)
        assert list(out.keys()) == ["rgb", "base_remap", "pts", "sigma", "dirs"]
        assert out["rgb"].shape == (4, 192, 3) and out["sigma"].shape == (4, 192) and out["base_remap"].shape == (4, 192, 256)
        errs = {"sigma": rel(out["sigma"], g[name + "_sigma"]), "rgb": rel(out["rgb"], g[name + "_rgb"]),
                "remap": rel(out["base_remap"][:, :48], g[name + "_remap_first48"])}
        print(precision, name, errs)
        for k, e in errs.items():
            assert e <= TOL[precision], (k, e)
            assert e <= TIGHT[precision], (k, e)
        # the encodings are exact to float32 rounding in both modes
        assert rel(out["pts"][:, :8], g[name + "_pts_enc_first8"]) <= 2e-7
        assert rel(out["dirs"][:, :8], g[name + "_dirs_enc_first8"]) <= 2e-7


@pytest.mark.parametrize("precision", ["fp16x3", "fp16mx", "fp16"])
def test_mlp_style_forward_encoded_inputs(golden, precision):
    """MLP_style.forward (models.py:95-117) on already-encoded float32 inputs."""
    g = golden("g4_nerf")
    sd = T(synth.nerf_state(0))
    ro, rd, ts = (torch.from_numpy(g[k]) for k in ("rays_o", "rays_d", "ts"))
    pts = ro[:, None, :] + ts[..., None].double() * rd[:, None, :]
    pe = fields.posenc(pts, 10).float()
    de = fields.posenc(rd[:, None, :].expand(-1, 192, -1), 4).float()
    m = make_nerf(0, "coarse", precision)
    out = m.net(pts=pe.cuda(), dirs=de.cuda())
    assert list(out.keys()) == ["rgb", "base_remap", "pts", "sigma"]
    assert rel(out["sigma"], g["coarse_mlp_sigma"]) <= TIGHT[precision]
    assert rel(out["rgb"], g["coarse_mlp_rgb"]) <= TIGHT[precision]
    ref = fields.nerf_mlp(sd, pe, de)
    assert rel(out["base_remap"], ref["base_remap"]) <= TIGHT[precision]


@pytest.mark.parametrize("M", [1, 15, 16, 17, 255, 256, 257, 1000])
def test_nerf_ragged_sizes(M):
    """Tail handling: sample counts around the 16-sample tile and the 128/256-sample workgroup."""
    rng = np.random.default_rng(M)
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (M, 3)))
    dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3)))
    ref = fields.style_nerf(T(synth.nerf_state(1)), pts, dirs)
    for precision in ("fp16x3", "fp16mx", "fp16"):
        out = make_nerf(1, "fine", precision)(pts=pts.cuda(), dirs=dirs.cuda())
        assert out["sigma"].shape == (M,)
        assert rel(out["sigma"], ref["sigma"]) <= TIGHT[precision] * 2
        assert rel(out["rgb"], ref["rgb"]) <= TIGHT[precision] * 2
        assert rel(out["base_remap"], ref["base_remap"]) <= TIGHT[precision] * 2


def test_nerf_empty_and_errors():
    from tgtc_style_amd import hip
    m = make_nerf(0, "coarse", "fp16x3")
    out = m(pts=torch.empty(0, 3, dtype=torch.float64).cuda(), dirs=torch.empty(0, 3, dtype=torch.float64).cuda())
    assert out["sigma"].shape == (0,)
    # a network the kernels are not built for is refused with an error, not silently mis-evaluated
    sd = synth.nerf_state(0, width=128)
    with pytest.raises(RuntimeError, match="built for"):
        hip.nerf_create({k: torch.from_numpy(v) for k, v in sd.items()}, "fp16x3")
    with pytest.raises(RuntimeError):
        m(pts=torch.zeros(4, 3, dtype=torch.float64), dirs=torch.zeros(4, 3, dtype=torch.float64))   # CPU tensors
    # the fp16+fp6 mode exists for the NeRF nets only; the style nets refuse it instead of falling back
    with pytest.raises(RuntimeError, match="NeRF nets only"):
        hip.style_create({k: torch.from_numpy(v) for k, v in synth.concat_state(0).items()},
                         {k: torch.from_numpy(v) for k, v in synth.style_state(0).items()}, "fp16mx")


@pytest.mark.parametrize("log2_range,bound", [(0, 5e-4), (3, 1e-3)])
def test_fp16mx_dynamic_range(log2_range, bound):
    """fp16mx corrections are block scaled (one exponent per weight row, one per 32 activations of a lane), so
    their benefit shrinks as the dynamic range inside a block grows; in the limit the mode degrades to plain
    fp16, never below it.  Rows of layers 1-3 are scaled by 2^U(-r, r) and the next layer's columns by the
    inverse (the function is unchanged, ReLU being positively homogeneous): at r = 3 (a 64 x spread between
    neighbouring features) the 1e-3 bar still holds on every output of the network."""
    sd = {k: v.copy() for k, v in synth.nerf_state(1).items()}
    rng = np.random.default_rng(5)
    for i in (1, 2, 3):
        s = (2.0 ** rng.integers(-log2_range, log2_range + 1, 256)).astype(np.float32)
        sd["net.base_layers.%d.weight" % i] *= s[:, None]
        sd["net.base_layers.%d.bias" % i] *= s
        sd["net.base_layers.%d.weight" % (i + 1)][:, -256:] /= s[None, :]
    pts = torch.from_numpy(rng.uniform(-1, 1, (512, 3)))
    dirs = torch.from_numpy(rng.uniform(-1, 1, (512, 3)))
    ref = fields.style_nerf(T(sd), pts, dirs)
    from tgtc_style_amd import models
    m = models.StyleNerf(type("A", (Args,), {"precision": "fp16mx"}), mode="fine")
    m.load_state_dict(T(sd))
    out = m.cuda()(pts=pts.cuda(), dirs=dirs.cuda())
    errs = {k: rel(out[k], ref[k]) for k in ("sigma", "rgb", "base_remap")}
    print("fp16mx, spread 2^+-%d:" % log2_range, errs)
    assert max(errs.values()) <= bound, errs


@pytest.mark.parametrize("M", [1, 31, 128, 129, 1000, 70001])
def test_two_tile_fp16mx_kernel_matches_the_one_tile_kernel_bit_for_bit(M):
    """tgtc_nerf_forward in fp16mx without the base_remap output runs the persistent two-tile kernel (csrc/mlp_nerf_mx2.hip: one
    generated instruction stream per pass, 128 samples per workgroup and pass, tools/gen_mx2_asm.py); with it, the one-tile
    kernel (mlp_nerf_mx.hip).  Same arithmetic per sample in the same order: the same bits, for any sample count (tails of a
    pass, of a tile, fewer passes than CUs, several passes per workgroup) and both input modes."""
    from tgtc_style_amd import hip
    lib = hip.load()
    rng = np.random.default_rng(M)
    pts = torch.from_numpy(rng.uniform(-1.5, 1.5, (M, 3))).cuda()
    dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3))).cuda()
    net = make_nerf(1, "fine", "fp16mx")
    h = net.packed().handle
    out = {}
    for tag in ("two_tile", "one_tile"):
        rgb = torch.full((M, 3), -7.0, device="cuda")
        sigma = torch.full((M,), -7.0, device="cuda")
        remap = torch.empty(M, 256, device="cuda") if tag == "one_tile" else None
        hip.check(lib.tgtc_nerf_forward(h, hip.ptr(pts), hip.ptr(dirs), M, hip.ptr(rgb), hip.ptr(sigma), hip.ptr(remap), None, None, hip.stream()))
        out[tag] = (rgb, sigma)
    assert torch.equal(out["two_tile"][0], out["one_tile"][0]) and torch.equal(out["two_tile"][1], out["one_tile"][1])
    ref = fields.style_nerf(T(synth.nerf_state(1)), pts.cpu(), dirs.cpu())
    assert rel(out["two_tile"][0].cpu(), ref["rgb"]) <= 1e-3 and rel(out["two_tile"][1].cpu(), ref["sigma"]) <= 1e-3
    # already-encoded inputs (MLP_style.forward's boundary): the same two kernels behind tgtc_nerf_mlp_forward
    pe, de = fields.posenc(pts.cpu(), 10).float().cuda().contiguous(), fields.posenc(dirs.cpu(), 4).float().cuda().contiguous()
    enc = {}
    for tag in ("two_tile", "one_tile"):
        rgb, sigma = torch.full((M, 3), -7.0, device="cuda"), torch.full((M,), -7.0, device="cuda")
        remap = torch.empty(M, 256, device="cuda") if tag == "one_tile" else None
        hip.check(lib.tgtc_nerf_mlp_forward(h, hip.ptr(pe), hip.ptr(de), M, hip.ptr(rgb), hip.ptr(sigma), hip.ptr(remap), hip.stream()))
        enc[tag] = (rgb, sigma)
    assert torch.equal(enc["two_tile"][0], enc["one_tile"][0]) and torch.equal(enc["two_tile"][1], enc["one_tile"][1])
    # the encodings as outputs (tgtc_nerf_forward's pts / dirs), written by the two-tile kernel's front end
    pe_o, de_o = torch.empty(M, 63, device="cuda"), torch.empty(M, 27, device="cuda")
    rgb, sigma = torch.empty(M, 3, device="cuda"), torch.empty(M, device="cuda")
    hip.check(lib.tgtc_nerf_forward(h, hip.ptr(pts), hip.ptr(dirs), M, hip.ptr(rgb), hip.ptr(sigma), None, hip.ptr(pe_o), hip.ptr(de_o), hip.stream()))
    assert torch.equal(rgb, out["two_tile"][0]) and float((pe_o.cpu() - pe.cpu()).abs().max()) <= 5e-5 and float((de_o.cpu() - de.cpu()).abs().max()) <= 5e-5
    # rays as input (the render chain's fine pass)
    R, N = max(M // 192, 1), 192
    ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3))).cuda()
    rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3))).cuda()
    ts = torch.sort(torch.from_numpy(rng.uniform(0, 1, (R, N)).astype(np.float32)).cuda(), -1).values.contiguous()
    rgb = torch.empty(R * N, 3, device="cuda")
    sigma = torch.empty(R * N, device="cuda")
    hip.check(lib.tgtc_nerf_forward_rays(h, hip.ptr(ro), hip.ptr(rd), hip.ptr(ts), R, N, hip.ptr(rgb), hip.ptr(sigma), hip.stream()))
    p = (ro[:, None, :] + ts.double()[:, :, None] * rd[:, None, :]).reshape(-1, 3)
    rgb2 = torch.empty(R * N, 3, device="cuda")
    sigma2 = torch.empty(R * N, device="cuda")
    remap = torch.empty(R * N, 256, device="cuda")
    hip.check(lib.tgtc_nerf_forward(h, hip.ptr(p.contiguous()), hip.ptr(rd[:, None, :].expand(R, N, 3).reshape(-1, 3).contiguous()), R * N,
                                    hip.ptr(rgb2), hip.ptr(sigma2), hip.ptr(remap), None, None, hip.stream()))
    # (positions: the kernel forms o + t d itself, possibly with one fused rounding less than torch: equal to fp16mx's own noise)
    assert rel(rgb, rgb2.cpu()) <= 2e-4 and rel(sigma, sigma2.cpu()) <= 2e-4


@pytest.mark.parametrize("R,N", [(1, 16), (40, 128), (257, 64), (2200, 128)])
def test_two_tile_fp16mx_sigma_kernel_matches_the_full_kernel_bit_for_bit(R, N):
    """fp16mx densities over rays (a coarse pass in fp16mx: `--precision fp16mx`): the sigma-only stream of the two-tile kernel (the
    first nine layers of the same weight stream) against the densities of the full one, itself pinned to the one-tile kernel above."""
    from tgtc_style_amd import hip
    lib = hip.load()
    rng = np.random.default_rng(R * 1000 + N + 7)
    ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3))).cuda()
    rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3))).cuda()
    ts = torch.sort(torch.from_numpy(rng.uniform(0, 1, (R, N)).astype(np.float32)).cuda(), -1).values.contiguous()
    net = make_nerf(0, "coarse", "fp16mx")      # (kept alive: the handle is the module's)
    h = net.packed().handle
    s_new = torch.full((R * N,), -7.0, device="cuda")
    hip.check(lib.tgtc_nerf_forward_rays(h, hip.ptr(ro), hip.ptr(rd), hip.ptr(ts), R, N, None, hip.ptr(s_new), hip.stream()))
    s_full, rgb = torch.full((R * N,), -7.0, device="cuda"), torch.empty(R * N, 3, device="cuda")
    hip.check(lib.tgtc_nerf_forward_rays(h, hip.ptr(ro), hip.ptr(rd), hip.ptr(ts), R, N, hip.ptr(rgb), hip.ptr(s_full), hip.stream()))
    assert torch.equal(s_new, s_full)


@pytest.mark.parametrize("R,N", [(1, 16), (3, 128), (40, 128), (700, 128), (257, 64), (2200, 128)])
def test_two_tile_fp16x3_sigma_kernel_matches_the_one_tile_kernel_bit_for_bit(R, N):
    """tgtc_nerf_forward_rays in fp16x3 with sigma as the only output (the coarse pass of a render) runs the persistent two-tile
    kernel (csrc/mlp_nerf_x3s.hip: one generated instruction stream per pass, tools/gen_x3_asm.py); asked for the colour too, the
    one-tile kernel (mlp_nerf.hip).  Same trunk arithmetic per sample in the same order: the same density bits for any ray and
    sample count (tails of a pass and of a tile, fewer passes than CUs, several passes per workgroup)."""
    from tgtc_style_amd import hip
    lib = hip.load()
    rng = np.random.default_rng(R * 1000 + N)
    ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3))).cuda()
    rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3))).cuda()
    ts = torch.sort(torch.from_numpy(rng.uniform(0, 1, (R, N)).astype(np.float32)).cuda(), -1).values.contiguous()
    net = make_nerf(0, "coarse", "fp16x3")
    h = net.packed().handle
    s_new = torch.full((R * N,), -7.0, device="cuda")
    hip.check(lib.tgtc_nerf_forward_rays(h, hip.ptr(ro), hip.ptr(rd), hip.ptr(ts), R, N, None, hip.ptr(s_new), hip.stream()))
    s_old, rgb = torch.full((R * N,), -7.0, device="cuda"), torch.empty(R * N, 3, device="cuda")
    hip.check(lib.tgtc_nerf_forward_rays(h, hip.ptr(ro), hip.ptr(rd), hip.ptr(ts), R, N, hip.ptr(rgb), hip.ptr(s_old), hip.stream()))
    assert torch.equal(s_new, s_old)
    pts = (ro[:, None, :] + ts.double()[:, :, None] * rd[:, None, :]).reshape(-1, 3).cpu()
    ref = fields.style_nerf(T(synth.nerf_state(0)), pts, rd[:, None, :].expand(R, N, 3).reshape(-1, 3).cpu())
    assert rel(s_new.cpu(), ref["sigma"]) <= 5e-5


def test_repack_on_weight_change():
    m = make_nerf(0, "coarse", "fp16x3")
    pts = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (64, 3))).cuda()
    a = m(pts=pts, dirs=pts)["sigma"].clone()
    m.load_state_dict({k: v for k, v in T(synth.nerf_state(1)).items()})
    b = m(pts=pts, dirs=pts)["sigma"]
    ref = fields.style_nerf(T(synth.nerf_state(1)), pts.cpu(), pts.cpu())["sigma"]
    assert not torch.allclose(a, b) and rel(b, ref) <= 5e-5


@pytest.mark.parametrize("precision", ["fp16x3", "fp16mx", "fp16"])
@pytest.mark.parametrize("nc,nf", [(128, 64), (64, 64)])
def test_render_rays_plain_golden(golden, precision, nc, nf):
    """The fused cal_geometry chain (rendering.py:27-51) against the reference's own render of 64 rays."""
    from tgtc_style_amd import rendering
    g = golden("g8_end_to_end")
    tag = "_%dc%df" % (nc, nf)
    ro, rd = torch.from_numpy(g["rays_o" + tag]).cuda(), torch.from_numpy(g["rays_d" + tag]).cuda()
    r = rendering.RayRenderer(make_nerf(0, "coarse", precision), make_nerf(1, "fine", precision))
    out = r.render(ro, rd, nc, nf, near=0., far=1., want_coarse=True)
    # Composited colours live in [0,1] and depths in [0,1]: absolute error == max-norm relative error.
    e_rgb = float((out["rgb"].cpu() - torch.from_numpy(g["plain_rgb" + tag])).abs().max())
    e_t = float((out["t"].cpu() - torch.from_numpy(g["plain_t" + tag])).abs().max())
    print(precision, tag, "rgb", e_rgb, "t", e_t)
    lim = {"fp16x3": 1e-3, "fp16mx": 1e-3, "fp16": 2e-2}[precision]
    # the composited depth rides on the inverse-CDF positions, which amplify sigma errors: fp16mx (sigma 1.5e-4) holds the
    # 1e-3 bar on colour but only 5e-3 on depth -- one reason fp16x3 stays the default parity mode
    lim_t = 5e-3 if precision == "fp16mx" else lim
    assert e_rgb <= lim and e_t <= lim_t
    # same chain assembled from the granular operators (what rendering.cal_geometry does) agrees with the fused call
    ref = fields.render_plain(T(synth.nerf_state(0)), T(synth.nerf_state(1)), ro.cpu(), rd.cpu(), nc, nf)
    assert float((out["rgb_coarse"].cpu() - ref["rgb_coarse"]).abs().max()) <= lim


def test_render_mixed_precision(golden):
    """Precision is a property of each network handle, so the two passes can differ.  The inverse-CDF step amplifies
    coarse-pass errors, the fine pass does not: coarse fp16x3 + fine fp16mx stays 3x inside the bar on colour AND depth
    (2.5e-4 / 2.1e-4 measured), while fp16mx in the coarse pass is what costs the depth its 2e-3."""
    from tgtc_style_amd import rendering
    g = golden("g8_end_to_end")
    ro, rd = torch.from_numpy(g["rays_o_128c64f"]).cuda(), torch.from_numpy(g["rays_d_128c64f"]).cuda()
    r = rendering.RayRenderer(make_nerf(0, "coarse", "fp16x3"), make_nerf(1, "fine", "fp16mx"))
    out = r.render(ro, rd, 128, 64)
    e_rgb = float((out["rgb"].cpu() - torch.from_numpy(g["plain_rgb_128c64f"])).abs().max())
    e_t = float((out["t"].cpu() - torch.from_numpy(g["plain_t_128c64f"])).abs().max())
    print("coarse fp16x3 + fine fp16mx: rgb", e_rgb, "t", e_t)
    assert e_rgb <= 1e-3 and e_t <= 1e-3


def test_render_adversarial_scene():
    """Stress scene (synth.nerf_state_adversarial): white-noise density, mostly empty space.  The
    coarse->fine chain is ill-conditioned there -- the fp32 reference itself moves by ~6e-4 when
    evaluated in fp64 (inverse-CDF positions scale with 1/pdf and pdf ~ 1e-5 in empty bins) -- so the
    end-to-end bound is 5e-3 while the per-stage outputs on identical points still meet 5e-5."""
    from tgtc_style_amd import models, rendering
    rng = np.random.default_rng(11)
    n = 128
    ro = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, (n, 2)), -np.ones((n, 1))], 1))
    rd = torch.from_numpy(np.concatenate([rng.uniform(-.3, .3, (n, 2)), 2 * np.ones((n, 1))], 1))
    sds = [T(synth.nerf_state_adversarial(0)), T(synth.nerf_state_adversarial(1))]
    nets = []
    for sd, mode in zip(sds, ("coarse", "fine")):
        m = models.StyleNerf(Args, mode=mode)
        m.load_state_dict(sd)
        nets.append(m.cuda())
    out = rendering.RayRenderer(*nets).render(ro.cuda(), rd.cuda(), 128, 64, want_coarse=True)
    ref = fields.render_plain(sds[0], sds[1], ro, rd, 128, 64)
    assert float((out["rgb_coarse"].cpu() - ref["rgb_coarse"]).abs().max()) <= 5e-5    # same points: tight
    assert float((out["rgb"].cpu() - ref["rgb_fine"]).abs().max()) <= 5e-3
    assert float((out["t"].cpu() - ref["t_fine"]).abs().max()) <= 5e-3
    pts, ts = raymarch.sample_coarse(ro, rd, 128, 0., 1.)
    per = nets[0](pts=pts.cuda(), dirs=rd[:, None, :].expand(-1, 128, -1).cuda())
    ref_c = fields.style_nerf(sds[0], pts, rd[:, None, :].expand(-1, 128, -1))
    assert rel(per["sigma"], ref_c["sigma"]) <= 5e-5 and rel(per["rgb"], ref_c["rgb"]) <= 5e-5


@pytest.mark.parametrize("precision", ["fp16x3", "fp16mx", "fp16"])
def test_render_full_size_properties(precision):
    """BASELINE size (128c+64f) on a whole 400-wide strip: properties that do not need the oracle."""
    from tgtc_style_amd import rendering, utils
    H, W = 400, 400
    focal = synth.fern_intrinsics(H, W)
    ro, rd = utils.gen_rays(H, W, focal, synth.spiral_pose(3), first_pixel=0, n=40 * W)
    r = rendering.RayRenderer(make_nerf(0, "coarse", precision), make_nerf(1, "fine", precision))
    a = r.render(ro, rd, 128, 64)
    assert a["rgb"].shape == (40 * W, 3) and bool(torch.isfinite(a["rgb"]).all())
    assert float(a["rgb"].min()) >= 0 and float(a["rgb"].max()) <= 1 + 1e-5       # convex combination of sigmoids
    assert float(a["t"].min()) >= 0 and float(a["t"].max()) <= 1 + 1e-5
    # rays are independent: any sub-range (a rank's shard, any chunking) reproduces the same bits
    b = r.render(ro[5000:9000].contiguous(), rd[5000:9000].contiguous(), 128, 64)
    assert torch.equal(a["rgb"][5000:9000], b["rgb"]) and torch.equal(a["t"][5000:9000], b["t"])
    if precision != "fp16x3":
        return
    # spot check 256 of the rays against the oracle
    idx = torch.arange(0, 40 * W, 40 * W // 256)[:256]
    ref = fields.render_plain(T(synth.nerf_state(0)), T(synth.nerf_state(1)), ro[idx].cpu(), rd[idx].cpu(), 128, 64)
    assert float((a["rgb"][idx].cpu() - ref["rgb_fine"]).abs().max()) <= 1e-3
