#!/usr/bin/env python3
"""Diagnostic (library built with -DTGTC_MX_TRACE): cycles between consecutive group starts inside layer 2, per wave."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import hip, synth, utils
lib = hip.load()
coarse, fine = bench.build_nets("fp16mx")
H = W = 400
o, d = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(0))
R, N = H * W, 192
ts = torch.sort(torch.rand(R, N, device="cuda"), -1)[0]
rgb = torch.empty(R, N, 3, device="cuda"); sig = torch.empty(R, N, device="cuda")
stamps = torch.zeros(64 * 8 * 32, dtype=torch.int64, device="cuda")
for it in range(3):
    hip.check(lib.tgtc_debug_set_stamps(hip.ptr(stamps) if it == 2 else None))
    hip.check(lib.tgtc_nerf_forward_rays(fine.packed().handle, hip.ptr(o), hip.ptr(d), hip.ptr(ts), R, N, hip.ptr(rgb), hip.ptr(sig), hip.stream()))
torch.cuda.synchronize()
hip.check(lib.tgtc_debug_set_stamps(None))
s = stamps.cpu().numpy().reshape(64, 8, 32).astype(np.int64)
for blk in (0, 17):
    print("block", blk)
    for w in range(8):
        t = s[blk, w, 14:32]
        print("  wave", w, "start %6d" % (t[0] - s[blk, :, 14].min()), " deltas", np.diff(t).tolist())
d = np.diff(s[:, :, 14:32], axis=2).reshape(-1, 17)
print("median deltas", np.median(d, 0).astype(int).tolist())
