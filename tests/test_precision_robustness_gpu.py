"""GPU: the headline precision (coarse fp16x3 + fine fp16mx, bench.py's default) END TO END on weights that do not look
like the seeded uniform nets every other test uses.

fp16mx's correction products are block scaled (one exponent per 32 activations a lane holds, one per weight row), so its
margin depends on the dynamic range inside a block -- and trained MLPs are heavier-tailed than U(+-1/sqrt(fan_in)).  This
sweep renders, for ten weight sets over four families (synth.heavy_tailed: the seeded base nets, log-normal per-feature
scales, x64 outlier features, log-normal per-element factors), the 64 rays of the reference's own end-to-end golden (g8)
plus a strip of a 400x400 frame (256 rays; 1 024 for the first set of every family) through the fused ray kernel and
compares with the fp32 oracle at the north-star 1e-3.  The worst margin is printed.  The mode survives the per-feature
families only because the packer equalises the ReLU trunk first (mlp_nerf_mx.hip, nerf_mx_equalise); with
TGTC_MX_NO_EQUALISE=1 this test fails on 'rows' and 'outliers' (CPU emulation: tests/probes/emu_mx_e2e.py,
profiles/r3_precision_emulation.md).  If any case here fails, bench.py's default goes back to fp16x3 in both passes.
"""
import numpy as np
import pytest
import torch

from oracle import fields, raymarch
from tgtc_style_amd import synth

pytestmark = pytest.mark.gpu

NC, NF = 128, 64
CASES = [("base", 0), ("base", 1), ("rows", 0), ("rows", 1), ("rows", 2), ("outliers", 0), ("outliers", 1), ("outliers", 2),
         ("elements", 0), ("elements", 1)]


def T(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


class Args:
    use_viewdir, act_type = True, "relu"
    embed_freq_coor, embed_freq_dir = 10, 4
    netdepth = netdepth_fine = 8
    netwidth = netwidth_fine = 256
    precision = "fp16x3"


def recalibrated(sd, base, ro, rd):
    """'elements' is a different function: re-centre its density head so that the scene keeps the base scene's density
    statistics (the family is about the numbers the kernels chew, not about an emptier or denser scene)."""
    pts, _ = raymarch.sample_coarse(ro[:48], rd[:48], 64, 0.0, 1.0)
    dirs = rd[:48, None, :].expand(-1, 64, -1)
    s0, s1 = fields.style_nerf(T(base), pts, dirs)["sigma"], fields.style_nerf(T(sd), pts, dirs)["sigma"]
    g = float(s0.std() / s1.std())
    sd["net.sigma_layer.weight"] = (sd["net.sigma_layer.weight"] * np.float32(g)).astype(np.float32)
    sd["net.sigma_layer.bias"] = (sd["net.sigma_layer.bias"] * np.float32(g) + np.float32(float(s0.mean()) - g * float(s1.mean()))).astype(np.float32)
    return sd


def weights(family, k, ro, rd):
    out = []
    for seed in (20 + 2 * k, 21 + 2 * k):                      # (coarse, fine): a fresh pair of nets per case
        base = synth.nerf_state(seed)
        sd = synth.heavy_tailed(base, seed, family)
        out.append(recalibrated(sd, base, ro, rd) if family == "elements" else sd)
    return out


def test_headline_precision_on_heavy_tailed_weights(golden):
    from tgtc_style_amd import models, rendering, utils
    g = golden("g8_end_to_end")
    H = W = 400
    fo, fd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(5))
    worst, seen = 0.0, set()
    for family, k in CASES:
        n_strip = 256 if family in seen else 1024
        seen.add(family)
        idx = torch.linspace(0, H * W - 1, n_strip).long().cuda()
        ro = torch.cat([torch.from_numpy(g["rays_o_128c64f"]).cuda(), fo[idx]]).contiguous()
        rd = torch.cat([torch.from_numpy(g["rays_d_128c64f"]).cuda(), fd[idx]]).contiguous()
        sc, sf = weights(family, k, ro.cpu(), rd.cpu())
        nets = []
        for sd, mode, prec in ((sc, "coarse", "fp16x3"), (sf, "fine", "fp16mx")):
            m = models.StyleNerf(type("A", (Args,), {"precision": prec}), mode=mode)
            m.load_state_dict(T(sd))
            nets.append(m.cuda())
        r = rendering.RayRenderer(*nets)
        assert r._fused_shape(NC, NF)
        out = r.render(ro, rd, NC, NF)
        rgb, t = out["rgb"].cpu(), out["t"].cpu()
        assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(t).all())
        ref = fields.render_plain(T(sc), T(sf), ro.cpu(), rd.cpu(), NC, NF)
        err_to = lambda o: torch.maximum((rgb - o["rgb_fine"]).abs().max(-1).values, (t - o["t_fine"]).abs())
        e = err_to(ref)
        # rays on a discontinuity of the reference algorithm (utils.py:367-369, :604-605): identified on the oracle, held to
        # the nearest branch (tests/test_whole_frame_gpu.py check_against_oracle)
        unstable, e_branch = torch.zeros_like(e, dtype=torch.bool), e.clone()
        for scale in (1.0 + 1e-7, 1.0 - 1e-7):
            moved = fields.render_plain(T(sc), T(sf), ro.cpu() * scale, rd.cpu(), NC, NF)
            unstable |= torch.maximum((moved["rgb_fine"] - ref["rgb_fine"]).abs().max(-1).values,
                                      (moved["t_fine"] - ref["t_fine"]).abs()) > 1e-4
            e_branch = torch.minimum(e_branch, err_to(moved))
        e_eff = torch.where(unstable, e_branch, e)
        case_max = float(e_eff.max())
        worst = max(worst, case_max)
        print("%-9s set %d, %4d rays: max %.2e (g8 rays %.2e), median %.2e, %d rays on a discontinuity" %
              (family, k, e.numel(), case_max, float(e_eff[:64].max()), float(e.median()), int(unstable.sum())))
        assert case_max <= 1e-3, (family, k)
        assert int(unstable.sum()) <= max(2, e.numel() // 100)
    print("worst case %.2e: margin %.1fx inside 1e-3" % (worst, 1e-3 / worst))
