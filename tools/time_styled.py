#!/usr/bin/env python3
"""Development timing: the stylised ray path, whole 400x400 frame, 128c+64f: the stylised ray kernel (one launch) against the
chain of per-sample kernels, same process, interleaved (TGTC_LIB selects a development build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import models, rendering, synth, utils
H = W = 400
o, d = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(0))
z = torch.randn(H * W, 32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for prec in sys.argv[1:] or ["fp16x3"]:
    r0 = bench.make_renderer(prec, True)
    for rep in range(2):
        for fused in (True, False):
            r = rendering.RayRenderer(r0.coarse, r0.fine, style=r0.style, fused=fused)
            for _ in range(2):
                out = r.render(o, d, 128, 64, z=z)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(3):
                out = r.render(o, d, 128, 64, z=z)
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 3
            print("styled %-8s %-6s %7.2f ms/frame  %8.0f rays/s  finite %s" % (
                prec, "fused" if fused else "chain", ms, H * W / ms * 1e3, bool(torch.isfinite(out["rgb"]).all())), flush=True)
