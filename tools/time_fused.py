#!/usr/bin/env python3
"""Development timing: the fused ray kernel vs the chain of per-sample kernels, whole 400x400 frame, 128c+64f."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import rendering, synth, utils
H = W = 400
o, d = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(0))
for prec in sys.argv[1:] or ["fp16x3", "fp16x3+fp16mx", "fp16"]:
    coarse, fine = bench.build_nets(prec)
    for fused in (True, False):
        r = rendering.RayRenderer(coarse, fine, fused=fused)
        for _ in range(2):
            out = r.render(o, d, 128, 64)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(4):
            out = r.render(o, d, 128, 64)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 4
        print("%-14s %-6s %7.2f ms/frame  %8.0f rays/s  whole-path frac %.3f  finite %s" % (
            prec, "fused" if fused else "chain", ms, H * W / ms * 1e3, H * W * bench.FLOP_PER_RAY / (ms * 1e-3) / 2.5e15,
            bool(torch.isfinite(out["rgb"]).all())), flush=True)
