#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused NeRF kernel from in-kernel s_memtime stamps.
Run on the GPU box: python tools/stamp_profile.py [fp16|fp16x3|fp16mx] [waves] [column tiles]

The "ideal MFMA" column prices a 16x16x32 MFMA at 16 cycles (the 2.4 GHz peak); in s_memtime units the pipe actually
turns one around in 12.6 (tools/microbench/lds_mfma_mix), so eff x 2 waves x 0.79 is the share of the matrix pipe."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgtc_style_amd import hip, synth, utils  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # waves per workgroup of the library under test
NCT = int(sys.argv[3]) if len(sys.argv) > 3 else (2 if prec == "fp16" else 1)
lib = hip.load()
coarse, fine = bench.build_nets(prec)
H = W = 400
o, d = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(0))
R, N = H * W, 192
ts = torch.sort(torch.rand(R, N, device="cuda"), -1)[0]
rgb = torch.empty(R, N, 3, device="cuda")
sig = torch.empty(R, N, device="cuda")
stamps = torch.zeros(64 * NW * 32, dtype=torch.int64, device="cuda")
for it in range(3):
    hip.check(lib.tgtc_debug_set_stamps(hip.ptr(stamps) if it == 2 else None))
    hip.check(lib.tgtc_nerf_forward_rays(fine.packed().handle, hip.ptr(o), hip.ptr(d), hip.ptr(ts), R, N, hip.ptr(rgb),
                                         hip.ptr(sig), hip.stream()))
torch.cuda.synchronize()
hip.check(lib.tgtc_debug_set_stamps(None))
s = stamps.cpu().numpy().reshape(64, NW, 32)[:, :, :14].astype(np.float64)
names = ["inputs", "glds issue", "posenc", "ring start", "L0", "L1", "L2", "L3", "L4", "L5(skip)", "L6", "L7",
         "sigma", "remap+rgb"]
dt = np.diff(s, axis=2)            # [64,4,13]
med = np.median(dt.reshape(-1, 13), 0)
tot = np.median(s[:, :, 13] - s[:, :, 0])
frags = [0, 0, 0, 0, 32, 128, 128, 128, 128, 160, 128, 128, 8, 204]
per = NCT * 16 * {"fp16": 1, "fp16x3": 3, "fp16mx": 1.5}[prec]
print("precision", prec, "median wave lifetime (stamped part): %.0f cycles" % tot)
for i in range(13):
    ideal = frags[i + 1] * per
    print("%-12s %8.0f cycles  %5.1f %%   ideal MFMA %6d  eff %s" % (names[i + 1], med[i], 100 * med[i] / tot, ideal,
                                                                     ("%.2f" % (ideal / med[i])) if ideal else "-"))
