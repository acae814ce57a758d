#!/usr/bin/env python3
"""Development timing of the fused training kernels of one network (131 072 samples): forward, backward (dgrad + wgrad)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import fused_train, models, synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (M, 3))).cuda()
dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3))).cuda()
g_rgb = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32) * 1e-3).cuda()
g_sig = torch.from_numpy(rng.standard_normal(M).astype(np.float32) * 1e-5).cuda()
net = models.StyleNerf(bench.NetArgs, mode="fine")
net.load_state_dict(bench.t_state(synth.nerf_state(1)))
net = net.cuda()
tr = fused_train.NerfTrainer()
params = [p.detach().float().contiguous() for p in fused_train.mlp_parameters(net.net)]
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n
rgb, sigma = tr.forward(params, pts, dirs)
print("M %d: forward %.3f ms  backward %.3f ms" % (M,
      timed(lambda: tr.forward(params, pts, dirs)), timed(lambda: tr.backward(params, rgb, g_rgb, g_sig))))
