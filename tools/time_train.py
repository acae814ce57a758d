#!/usr/bin/env python3
"""Development timing: one Origin_train iteration (training.origin_train_step) on the fused training kernels (TGTC_TRAIN_UNFUSED=1: the per-layer HIP dense layers),
1024 rays x (64 coarse + 128 fine-pass samples)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import models, synth, training
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3))).cuda()
rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1.0]).cuda()
gt = torch.from_numpy(rng.uniform(0.2, 0.8, (R, 3)).astype(np.float32)).cuda()
m, mf = models.StyleNerf(bench.NetArgs, mode="coarse"), models.StyleNerf(bench.NetArgs, mode="fine")
m.load_state_dict(bench.t_state(synth.nerf_state(0))), mf.load_state_dict(bench.t_state(synth.nerf_state(1)))
FUSED = os.environ.get("TGTC_TRAIN_UNFUSED") != "1"
m, mf = m.cuda().trainable(fused=FUSED), mf.cuda().trainable(fused=FUSED)
opt = torch.optim.Adam(list(m.parameters()) + list(mf.parameters()), lr=5e-4)
for i in range(13):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.time()
    r = training.origin_train_step(m, mf, opt, ro, rd, gt, 64, 64, 0., 1., sigma_noise_std=0.1, as_float=False)
t_cpu = (time.time() - t0) / 10          # host time to ENQUEUE an iteration (the losses come back as device scalars, unread)
torch.cuda.synchronize()
dt = (time.time() - t0) / 10
samples = R * (64 + 128)
print("host enqueue time per iteration %.2f ms" % (t_cpu * 1e3))
print(("fused " if FUSED else "unfused ") + "origin_train_step: %d rays, %d network samples: %.1f ms per iteration (%.2f M samples/s), loss %.4f" % (R, samples, dt * 1e3, samples / dt / 1e6, float(r["loss"])))
