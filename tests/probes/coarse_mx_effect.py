#!/usr/bin/env python3
"""What would a coarse pass in fp16mx change?  Whole 400x400 frames (fern-shaped synthetic scene, 128c + 64f) rendered with coarse
fp16x3 + fine fp16mx (the headline pair) and with fp16mx in both passes; per-ray differences of colour and depth.  GPU only; no oracle:
the question is how many rays a certified hybrid would have to send back through fp16x3."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tgtc_style_amd import rendering, synth, utils

H = W = 400
focal = synth.fern_intrinsics(H, W)
ref = bench.make_renderer("fp16x3+fp16mx", False)
alt = bench.make_renderer("fp16mx", False)
strict = bench.make_renderer("fp16x3", False)
for pose in (0, 40, 80):
    o, d = utils.gen_rays(H, W, focal, synth.spiral_pose(pose))
    a, b, c = (r.render(o, d, 128, 64, near=0., far=1.) for r in (ref, alt, strict))
    for name, x, y in (("coarse fp16mx vs coarse fp16x3 (fine fp16mx both)", b, a), ("headline pair vs strict fp16x3", a, c)):
        e_rgb = (x["rgb"] - y["rgb"]).abs().amax(-1)
        e_t = (x["t"] - y["t"]).abs()
        q = lambda v, p: float(torch.quantile(v.float(), p))
        print("pose %3d  %-52s rgb: max %.2e p99.9 %.2e  >1e-3: %6.3f %%   depth: max %.2e p99.9 %.2e  >1e-3: %6.3f %%" % (
            pose, name, float(e_rgb.max()), q(e_rgb, 0.999), 100 * float((e_rgb > 1e-3).float().mean()),
            float(e_t.max()), q(e_t, 0.999), 100 * float((e_t > 1e-3).float().mean())), flush=True)
