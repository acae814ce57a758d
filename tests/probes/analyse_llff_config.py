#!/usr/bin/env python3
"""CPU side of the dissection started by dump_llff_config.py: which rays of the 40x40 frame disagree with the oracle, at which
stage the disagreement appears, and whether the reference itself is discontinuous there.

    python tests/probes/analyse_llff_config.py flower [gpurun_out/dissect_flower.npz]

For every precision pair of the dump: distance of the fused output to the float32 oracle; for the worst rays, stage by stage:
coarse sigma / weights (HIP vs oracle float32 and float64), the merged fine depths (HIP sampler on the HIP weights vs the
oracle's sampler on the oracle's weights AND vs the oracle's sampler on the HIP weights: the last one separates "the sampler
differs" from "the sampler amplifies a weight difference"), fine sigma / rgb on the HIP depths, the composited result."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fields, raymarch
from tgtc_style_amd import config as cfg, synth

scene = sys.argv[1] if len(sys.argv) > 1 else "flower"
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "dissect_%s.npz" % scene)
D = np.load(path)
seed = {"fern": 0, "flower": 30, "horns": 32, "orchids": 34, "trex": 36}[scene]
args = cfg.parse_args(["--config", os.path.join(ROOT, "configs", scene + ".txt")])
nc, nf = args.N_samples, args.N_samples_fine
t = lambda sd, dt: {k: torch.from_numpy(np.ascontiguousarray(v)).to(dt) for k, v in sd.items()}
sds = [synth.nerf_state(seed), synth.nerf_state(seed + 1)]
ro, rd = torch.from_numpy(D["rays_o"]), torch.from_numpy(D["rays_d"])
f32, f64 = torch.float32, torch.float64
T = lambda k: torch.from_numpy(D[k])


def oracle(dc, df, o=ro, d=rd):
    return fields.render_plain(t(sds[0], dc), t(sds[1], df), o, d, nc, nf, dtype=dc, dtype_fine=df)


o32, o64 = oracle(f32, f32), oracle(f64, f64)
vec = lambda o: torch.cat([o["rgb_fine"].float(), o["t_fine"].float()[:, None]], 1)
spread = (vec(o32) - vec(o64)).abs().max(1).values
print("%s %dc+%df, %d rays; float32 vs float64 oracle: max %.2e, %d rays > 1e-4, %d > 1e-3" % (
    scene, nc, nf, ro.shape[0], float(spread.max()), int((spread > 1e-4).sum()), int((spread > 1e-3).sum())))
for tag in ("x3", "mx"):
    x = torch.cat([T("fused_rgb_" + tag), T("fused_t_" + tag)[:, None]], 1)
    e32 = (x - vec(o32)).abs().max(1).values
    e64 = (x - vec(o64)).abs().max(1).values
    emin = torch.minimum(e32, e64)
    print("\n== fine %s: |HIP - oracle32| max %.2e (%d rays > 1e-3, %d > 1e-4), p99 %.2e; min(f32, f64 oracle) max %.2e (%d rays > 1e-3)" % (
        tag, float(e32.max()), int((e32 > 1e-3).sum()), int((e32 > 1e-4).sum()), float(torch.quantile(e32, 0.99)), float(emin.max()), int((emin > 1e-3).sum())))
    worst = torch.argsort(e32, descending=True)[:5]
    # stage comparison on the worst rays
    sig_c, w_c, ts_f = T("sigma_c_" + tag), T("w_c_" + tag), T("ts_f_" + tag)
    pts, ts = raymarch.sample_coarse(ro, rd, nc, 0., 1., dtype=f32)
    dirs = rd[:, None, :].expand(-1, nc, -1)
    c32 = fields.style_nerf(t(sds[0], f32), pts, dirs, dtype=f32)
    c64 = fields.style_nerf(t(sds[0], f64), pts.double(), dirs, dtype=f64)
    _, _, w32 = raymarch.composite(c32["rgb"], c32["sigma"], ts)
    _, _, w64 = raymarch.composite(c64["rgb"], c64["sigma"], ts.double())
    _, tsf_or_on_hipw = raymarch.sample_fine(ro, rd, ts, w_c, nf)                # oracle sampler on the HIP weights
    for r in worst.tolist():
        print(" ray %4d: err32 %.2e err64 %.2e  f32-vs-f64 oracle spread %.2e" % (r, float(e32[r]), float(e64[r]), float(spread[r])))
        ds = (sig_c[r] - c32["sigma"][r]).abs().max() / c32["sigma"][r].abs().max()
        ds64 = (c64["sigma"][r].float() - c32["sigma"][r]).abs().max() / c32["sigma"][r].abs().max()
        dw = (w_c[r] - w32[r]).abs().max()
        dw64 = (w64[r].float() - w32[r]).abs().max()
        print("    coarse sigma: HIP vs f32 %.2e (max-norm rel), f64 vs f32 %.2e; weights: HIP vs f32 %.2e, f64 vs f32 %.2e" % (
            float(ds), float(ds64), float(dw), float(dw64)))
        dts = (ts_f[r] - o32["ts_fine"][r]).abs()
        dts_same_w = (ts_f[r] - tsf_or_on_hipw[r]).abs()
        dts64 = (o64["ts_fine"][r].float() - o32["ts_fine"][r]).abs()
        print("    fine depths: HIP vs oracle32 max %.2e (%d samples > 1e-4); HIP sampler vs ORACLE sampler on the SAME (HIP) weights max %.2e; "
              "oracle64 vs oracle32 max %.2e (%d > 1e-4)" % (float(dts.max()), int((dts > 1e-4).sum()), float(dts_same_w.max()),
                                                             float(dts64.max()), int((dts64 > 1e-4).sum())))
        # the cdf steps around the moved samples
        w = w32[r].double()
        pdf = (w[1:-1] + 1e-5) / (w[1:-1] + 1e-5).sum()
        cdf = torch.cat([torch.zeros(1, dtype=f64), torch.cumsum(pdf, 0)])
        steps = cdf[1:] - cdf[:-1]
        print("    cdf steps of this ray: min %.2e, %d steps in [0.5e-5, 2e-5] (the interpolation threshold of utils.py:604-605 is 1e-5)" % (
            float(steps.min()), int(((steps > 0.5e-5) & (steps < 2e-5)).sum())))
        # fine network on the HIP depths: is the fine pass itself accurate?
        p = ro[r][None, None, :] + rd[r][None, None, :] * ts_f[r].double()[None, :, None]
        f = fields.style_nerf(t(sds[1], f32), p.float().double(), rd[r][None, None, :].expand(-1, nc + nf, -1), dtype=f32)
        sf = T("sigma_f_" + tag)[r]
        print("    fine net on the HIP depths: sigma HIP vs oracle32 %.2e (max-norm rel), rgb %.2e" % (
            float((sf - f["sigma"][0]).abs().max() / f["sigma"][0].abs().max()),
            float((T("rgb_pts_f_" + tag)[r] - f["rgb"][0]).abs().max())))
        rgbo, to, _ = raymarch.composite(f["rgb"], f["sigma"], ts_f[r][None])
        xo = torch.cat([rgbo[0], to])
        print("    oracle32 fine net + compositing on the HIP depths vs HIP output: %.2e   <- what is left when the sampler's branch is taken as given" % (
            float((xo - x[r]).abs().max())))
