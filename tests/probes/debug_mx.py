#!/usr/bin/env python3
"""Development probe: fp16mx NeRF kernels (all input modes) vs the oracle, with NaN maps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import fields
from tgtc_style_amd import hip, synth
lib = hip.load()
PREC = os.environ.get("PREC", "fp16mx")
coarse, fine = bench.build_nets(PREC)
rng = np.random.default_rng(0)
M = int(os.environ.get("M", "1000"))
pts = torch.from_numpy(rng.uniform(-1, 1, (M, 3)))
dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3)))
out = fine(pts=pts.cuda(), dirs=dirs.cuda())
torch.cuda.synchronize()
t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
ref = fields.style_nerf(t(synth.nerf_state(1)), pts, dirs)
for k in ("sigma", "base_remap", "rgb"):
    a, b = out[k].cpu().double().reshape(M, -1), ref[k].double().reshape(M, -1)
    nan = torch.isnan(a)
    err = (a - b).abs()
    err[nan] = 0
    print(k, "nan rows", int(nan.any(1).sum()), "of", M, " nan cols", sorted(set(torch.nonzero(nan)[:, 1].tolist()))[:20],
          " max rel err (non-nan) %.3e" % float(err.max() / b.abs().max()), " rms rel %.3e" % float(err.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()))
    if k == "base_remap" and not nan.any():
        e = err / b.abs().max()
        print("   worst columns", torch.topk(e.max(0).values, 8))

# ray mode, sigma-only and full
R, N = 37, 192
ro = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1))
rd = torch.from_numpy(np.concatenate([rng.uniform(-.3, .3, (R, 2)), 2 * np.ones((R, 1))], 1))
ts = torch.from_numpy(np.sort(rng.uniform(0, 1, (R, N)).astype(np.float32), -1))
d_ro, d_rd, d_ts = ro.cuda(), rd.cuda(), ts.cuda()
rgb = torch.zeros(R, N, 3, device="cuda"); sig = torch.zeros(R, N, device="cuda"); sig2 = torch.zeros(R, N, device="cuda")
hip.check(lib.tgtc_nerf_forward_rays(fine.packed().handle, hip.ptr(d_ro), hip.ptr(d_rd), hip.ptr(d_ts), R, N, hip.ptr(rgb), hip.ptr(sig), hip.stream()))
hip.check(lib.tgtc_nerf_forward_rays(fine.packed().handle, hip.ptr(d_ro), hip.ptr(d_rd), hip.ptr(d_ts), R, N, None, hip.ptr(sig2), hip.stream()))
torch.cuda.synchronize()
p2 = ro[:, None, :] + ts[..., None].double() * rd[:, None, :]
ref2 = fields.style_nerf(t(synth.nerf_state(1)), p2, rd[:, None, :].expand(-1, N, -1))
for name, a, b in (("sigma full", sig, ref2["sigma"]), ("sigma only", sig2, ref2["sigma"]), ("rgb", rgb, ref2["rgb"])):
    a = a.cpu().double().reshape(R * N, -1); b = b.double().reshape(R * N, -1)
    nan = torch.isnan(a).any(1)
    idx = torch.nonzero(nan)[:, 0]
    err = (a - b).abs(); err[torch.isnan(err)] = 0
    bad = torch.nonzero(err.max(1).values > 1e-3 * b.abs().max())[:, 0]
    print(name, "nan samples", int(nan.sum()), "of", R * N, "first", idx[:12].tolist(), "| bad samples", len(bad), "first", bad[:12].tolist(),
          "lanes(n)", sorted(set((bad % 16).tolist()))[:16], "waves", sorted(set(((bad // 16) % 8).tolist())), " max rel %.2e" % float(err.max() / b.abs().max()))
