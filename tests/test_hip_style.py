"""GPU parity of the stylised path: latents -> concat MLP -> style MLP, granular and fused, through the C ABI.
Tolerances as in test_hip_nerf.py (max-norm relative; 1e-3 north-star, ~1e-5 achieved by fp16x3)."""
import numpy as np
import pytest
import torch

from oracle import fields
from tgtc_style_amd import synth

pytestmark = pytest.mark.gpu
TIGHT = {"fp16x3": 5e-5, "fp16": 1e-2}


def T(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def rel(a, ref):
    a, ref = torch.as_tensor(a).double().cpu(), torch.as_tensor(np.asarray(ref)).double()
    return float((a - ref).abs().max() / ref.abs().max())


class Args:
    use_viewdir, act_type = True, "relu"
    embed_freq_coor, embed_freq_dir = 10, 4
    netdepth = netdepth_fine = 8
    netwidth = netwidth_fine = 256
    style_D, vae_latent = 8, 32
    precision = "fp16x3"


def make(precision):
    from tgtc_style_amd import models
    a = type("A", (Args,), {"precision": precision})
    cm = models.StyleMLP_before_concat(a)
    cm.load_state_dict(T(synth.concat_state(2)))
    sm = models.StyleMLP_Wild_multilayers(a)
    sm.load_state_dict(T(synth.style_state(3)))
    nets = []
    for seed, mode in ((0, "coarse"), (1, "fine")):
        m = models.StyleNerf(a, mode=mode)
        m.load_state_dict(T(synth.nerf_state(seed)))
        nets.append(m.cuda())
    return cm.cuda(), sm.cuda(), nets


@pytest.mark.parametrize("precision", ["fp16x3", "fp16"])
def test_style_mlps_golden(golden, precision):
    """StyleMLP_before_concat.forward / StyleMLP_Wild_multilayers.forward on the reference's outputs (g7)."""
    g = golden("g7_style")
    cm, sm, _ = make(precision)
    x, z, conc = (torch.from_numpy(g[k]).cuda() for k in ("x", "z", "conc"))
    cf = cm(x=x, latent=z)["concat_features"]
    rgb = sm(x=x, concated=conc, latent=z)["rgb"]
    assert cf.shape == (24, 256) and rgb.shape == (24, 3)
    e1, e2 = rel(cf, g["concat_features"]), rel(rgb, g["style_rgb"])
    print(precision, "concat", e1, "style", e2)
    assert e1 <= TIGHT[precision] and e2 <= TIGHT[precision]
    # state-dict key names are the reference's
    assert list(cm.state_dict().keys())[:2] == ["layers.0.weight", "layers.0.bias"] and len(cm.layers) == 5
    assert len(sm.layers) == 8 and tuple(sm.layers[7].weight.shape) == (3, 288)


@pytest.mark.parametrize("M", [1, 33, 128, 300])
def test_style_mlps_ragged(M):
    rng = np.random.default_rng(M)
    x = torch.from_numpy(rng.uniform(-1, 1, (M, 63)).astype(np.float32))
    z = torch.from_numpy(rng.standard_normal((M, 32)).astype(np.float32))
    conc = torch.from_numpy(np.maximum(rng.standard_normal((M, 512)), 0).astype(np.float32))
    ref_c = fields.concat_mlp(T(synth.concat_state(2)), x, z)["concat_features"]
    ref_s = fields.style_mlp(T(synth.style_state(3)), x, conc, z)["rgb"]
    for precision in ("fp16x3", "fp16"):
        cm, sm, _ = make(precision)
        assert rel(cm(x=x.cuda(), latent=z.cuda())["concat_features"], ref_c) <= TIGHT[precision] * 2
        assert rel(sm(x=x.cuda(), concated=conc.cuda(), latent=z.cuda())["rgb"], ref_s) <= TIGHT[precision] * 2


@pytest.mark.parametrize("precision", ["fp16x3", "fp16"])
def test_styled_forward_rays_vs_oracle(precision):
    """The fused 24-layer kernel on identical points: sigma and stylised rgb per sample (rendering.py:122-142)."""
    from tgtc_style_amd import hip, models
    cm, sm, nets = make(precision)
    pair = models.StylePair(cm, sm)
    rng = np.random.default_rng(5)
    R, N = 7, 192
    ro = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1))
    rd = torch.from_numpy(np.concatenate([rng.uniform(-.3, .3, (R, 2)), 2 * np.ones((R, 1))], 1))
    ts = torch.from_numpy(np.sort(rng.uniform(0, 1, (R, N)).astype(np.float32), -1))
    z = torch.from_numpy(rng.standard_normal((R, 32)).astype(np.float32))
    rgb = torch.empty(R, N, 3, device="cuda")
    sigma = torch.empty(R, N, device="cuda")
    lib = hip.load()
    d_ro, d_rd, d_ts, d_z = ro.cuda(), rd.cuda(), ts.cuda(), z.cuda()   # keep alive across the async launch
    hip.check(lib.tgtc_styled_forward_rays(nets[1].packed().handle, pair.packed().handle, hip.ptr(d_ro),
                                           hip.ptr(d_rd), hip.ptr(d_ts), hip.ptr(d_z), R, N,
                                           hip.ptr(rgb), hip.ptr(sigma), hip.stream()))
    torch.cuda.synchronize()
    pts = ro[:, None, :] + ts[..., None].double() * rd[:, None, :]
    ref_rgb, ref_sig = fields._styled_pass(T(synth.nerf_state(1)), T(synth.concat_state(2)), T(synth.style_state(3)),
                                           pts, rd[:, None, :].expand(-1, N, -1), z)
    e1, e2 = rel(sigma, ref_sig), rel(rgb, ref_rgb)
    print(precision, "styled sigma", e1, "rgb", e2)
    assert e1 <= TIGHT[precision] and e2 <= TIGHT[precision]


@pytest.mark.parametrize("precision", ["fp16x3", "fp16"])
@pytest.mark.parametrize("nc,nf", [(128, 64), (64, 64)])
def test_render_rays_styled_golden(golden, precision, nc, nf):
    """The render_style chain (rendering.py:118-178) on the reference's own stylised render of 64 rays, with and
    without the stratified jitter."""
    from tgtc_style_amd import models, rendering
    g = golden("g8_end_to_end")
    tag = "_%dc%df" % (nc, nf)
    cm, sm, nets = make(precision)
    lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    lat.load_state_dict(T(synth.latents_state(4)))
    lat = lat.cuda()
    lat.sigma_scale = 1.0
    ro, rd = torch.from_numpy(g["rays_o" + tag]).cuda(), torch.from_numpy(g["rays_d" + tag]).cuda()
    R = ro.shape[0]
    z = lat(style_ids=torch.zeros(R, dtype=torch.long), frame_ids=torch.full((R,), 33, dtype=torch.long), type="llff")
    r = rendering.RayRenderer(nets[0], nets[1], models.StylePair(cm, sm))
    lim = {"fp16x3": 1e-3, "fp16": 2e-2}[precision]
    for jt, jit in (("", None), ("_jit", torch.from_numpy(g["jit" + tag]).cuda())):
        out = r.render(ro, rd, nc, nf, near=0., far=1., jitter=jit, z=z, want_coarse=True)
        e = {k: float((out[a].cpu() - torch.from_numpy(g[b + jt + tag])).abs().max())
             for k, a, b in (("rgb", "rgb", "styled_rgb"), ("t", "t", "styled_t"), ("rgb_coarse", "rgb_coarse", "styled_rgb_coarse"))}
        print(precision, tag, jt, e)
        assert max(e.values()) <= lim, e


def test_render_styled_full_size_properties():
    """BASELINE size (128c+64f) on a 20-row strip of a 400-wide frame through the fused stylised chain: range,
    finiteness, shard independence (a rank's sub-range reproduces the same bits), and a spot check of 128 rays against
    the oracle."""
    from oracle import fields
    from tgtc_style_amd import models, rendering, utils
    H, W = 400, 400
    cm, sm, nets = make("fp16x3")
    lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    lat.load_state_dict(T(synth.latents_state(4)))
    lat = lat.cuda()
    lat.sigma_scale = 1.0
    ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(5), first_pixel=180 * W, n=20 * W)
    R = ro.shape[0]
    z = lat(style_ids=torch.zeros(R, dtype=torch.long), frame_ids=torch.full((R,), 7, dtype=torch.long), type="llff")
    r = rendering.RayRenderer(nets[0], nets[1], models.StylePair(cm, sm))
    a = r.render(ro, rd, 128, 64, z=z)
    assert a["rgb"].shape == (R, 3) and bool(torch.isfinite(a["rgb"]).all()) and bool(torch.isfinite(a["t"]).all())
    assert float(a["rgb"].min()) >= 0 and float(a["rgb"].max()) <= 1 + 1e-5
    b = r.render(ro[3000:5000].contiguous(), rd[3000:5000].contiguous(), 128, 64, z=z[3000:5000].contiguous())
    assert torch.equal(a["rgb"][3000:5000], b["rgb"]) and torch.equal(a["t"][3000:5000], b["t"])
    idx = torch.arange(0, R, R // 128)[:128]
    ref = fields.render_styled(T(synth.nerf_state(0)), T(synth.nerf_state(1)), T(synth.concat_state(2)), T(synth.style_state(3)),
                               ro[idx].cpu(), rd[idx].cpu(), z[idx].cpu(), 128, 64)
    e = float((a["rgb"][idx].cpu() - ref["rgb_fine"]).abs().max())
    print("styled 128c+64f, 128 rays vs oracle: rgb", e)
    assert e <= 1e-3
