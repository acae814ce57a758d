#!/usr/bin/env python3
"""Correctness probe for development libraries (TGTC_LIB=...): fp16 ray-mode NeRF kernels vs the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import fields
from tgtc_style_amd import hip, synth
lib = hip.load()
PREC = os.environ.get("PREC", "fp16")
TOL = {"fp16": 1e-2, "fp16x3": 5e-5, "fp16mx": 1e-3}[PREC]
coarse, fine = bench.build_nets(PREC)
rng = np.random.default_rng(0)
R, N = 301, 192
ro = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1))
rd = torch.from_numpy(np.concatenate([rng.uniform(-.3, .3, (R, 2)), 2 * np.ones((R, 1))], 1))
ts = torch.from_numpy(np.sort(rng.uniform(0, 1, (R, N)).astype(np.float32), -1))
d_ro, d_rd, d_ts = ro.cuda(), rd.cuda(), ts.cuda()
rgb = torch.zeros(R, N, 3, device="cuda"); sig = torch.zeros(R, N, device="cuda"); sig2 = torch.zeros(R, N, device="cuda")
hip.check(lib.tgtc_nerf_forward_rays(fine.packed().handle, hip.ptr(d_ro), hip.ptr(d_rd), hip.ptr(d_ts), R, N, hip.ptr(rgb), hip.ptr(sig), hip.stream()))
hip.check(lib.tgtc_nerf_forward_rays(fine.packed().handle, hip.ptr(d_ro), hip.ptr(d_rd), hip.ptr(d_ts), R, N, None, hip.ptr(sig2), hip.stream()))
torch.cuda.synchronize()
t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
pts = ro[:, None, :] + ts[..., None].double() * rd[:, None, :]
ref = fields.style_nerf(t(synth.nerf_state(1)), pts, rd[:, None, :].expand(-1, N, -1))
rel = lambda a, b: float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max())
print("sigma(full) %.2e  sigma(sigma-only) %.2e  rgb %.2e" % (rel(sig, ref["sigma"]), rel(sig2, ref["sigma"]), rel(rgb, ref["rgb"])))
assert rel(sig, ref["sigma"]) < TOL and rel(sig2, ref["sigma"]) < TOL and rel(rgb, ref["rgb"]) < TOL
print("OK")
