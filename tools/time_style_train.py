#!/usr/bin/env python3
"""Development timing: one Style_train iteration (training.style_train_step: frozen NeRF nets on the fused inference kernels,
the concat / style MLPs and the latent table trained through the per-layer differentiable HIP layers), 1024 rays x (64 + 128)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import models, synth, training
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(2)
ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3))).cuda()
rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1.0]).cuda()
gt = torch.from_numpy(rng.uniform(0.2, 0.8, (R, 3)).astype(np.float32)).cuda()
sid = torch.zeros(R, dtype=torch.long).cuda()
fid = torch.from_numpy(rng.integers(0, 20, R)).cuda()
A = type("A", (bench.NetArgs,), {"style_D": 8, "vae_latent": 32})
model, model_fine = models.StyleNerf(A, mode="coarse"), models.StyleNerf(A, mode="fine")
model.load_state_dict(bench.t_state(synth.nerf_state(0))), model_fine.load_state_dict(bench.t_state(synth.nerf_state(1)))
model, model_fine = model.cuda(), model_fine.cuda()
model.set_enable_style(True), model_fine.set_enable_style(True)
cm, sm = models.StyleMLP_before_concat(A), models.StyleMLP_Wild_multilayers(A)
cm.load_state_dict(bench.t_state(synth.concat_state(2))), sm.load_state_dict(bench.t_state(synth.style_state(3)))
lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
lat.load_state_dict(bench.t_state(synth.latents_state(4, style_num=1, frame_num=20)))
cm, sm, lat = cm.cuda().trainable(), sm.cuda().trainable(), lat.cuda().trainable()
lat.sigma_scale = 1.0
opt = torch.optim.Adam(list(cm.parameters()) + list(sm.parameters()) + [lat.latents], lr=1e-3)
for i in range(13):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.time()
    r = training.style_train_step(model, model_fine, cm, sm, lat, opt, ro, rd, gt, sid, fid, 64, 64, 0., 1., sigma_noise_std=0.1,
                                  logp_loss_lambda=1e-3, as_float=False)
t_cpu = (time.time() - t0) / 10
torch.cuda.synchronize()
dt = (time.time() - t0) / 10
print("host enqueue %.2f ms; style_train_step: %d rays, %d network samples: %.1f ms per iteration, loss %.4f" % (
    t_cpu * 1e3, R, R * 192, dt * 1e3, float(r["loss"])))
