#!/usr/bin/env python3
"""GPU side of the dissection of one LLFF config of test_every_llff_config_inside_1e3_at_the_headline_precision (round 3's
red flower ray): renders the test's 40x40 frame with the fused ray kernel in fp16x3+fp16x3 and fp16x3+fp16mx and, stage by
stage through the per-stage operators, the coarse depths / sigma / weights, the merged fine depths and the fine network's
sigma / rgb in both fine precisions.  Everything goes to gpurun_out/dissect_<scene>.npz; tests/probes/analyse_llff_config.py
compares it with the oracle on the CPU.

    python tests/probes/dump_llff_config.py flower"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tgtc_style_amd import config as cfg, models, rendering, synth, utils

scene = sys.argv[1] if len(sys.argv) > 1 else "flower"
seed = {"fern": 0, "flower": 30, "horns": 32, "orchids": 34, "trex": 36}[scene]
t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
sds = [synth.nerf_state(seed), synth.nerf_state(seed + 1)]


def nets(prec):
    args = cfg.parse_args(["--config", os.path.join(ROOT, "configs", scene + ".txt"), "--precision", prec])
    out = []
    for sd, mode in zip(sds, ("coarse", "fine")):
        m = models.StyleNerf(args, mode=mode)
        m.load_state_dict(t(sd))
        out.append(m.cuda())
    return args, out


H = W = 40
ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(seed + 3))
dump = {"rays_o": ro.cpu().numpy(), "rays_d": rd.cpu().numpy()}
for prec in ("fp16x3", "fp16x3+fp16mx"):
    args, (coarse, fine) = nets(prec)
    tag = "x3" if prec == "fp16x3" else "mx"
    nc, nf = args.N_samples, args.N_samples_fine
    out = rendering.RayRenderer(coarse, fine).render(ro, rd, nc, nf)
    dump["fused_rgb_" + tag], dump["fused_t_" + tag] = out["rgb"].cpu().numpy(), out["t"].cpu().numpy()
    # stage by stage (the chain of per-stage operators: reference rendering.py:27-51)
    pts, ts = utils.sampling_pts_uniform(ro, rd, nc, near=0., far=1.)
    c = coarse(pts=pts, dirs=rd[:, None, :])
    rgb_c, t_c, w_c = utils.alpha_composition(c["rgb"], c["sigma"], ts)
    pts_f, ts_f = utils.sampling_pts_fine_torch(ro, rd, ts, w_c, nf)
    f = fine(pts=pts_f, dirs=rd[:, None, :])
    rgb_f, t_f, w_f = utils.alpha_composition(f["rgb"], f["sigma"], ts_f)
    for k, v in (("ts_c", ts), ("sigma_c", c["sigma"]), ("w_c", w_c), ("ts_f", ts_f), ("sigma_f", f["sigma"]), ("rgb_pts_f", f["rgb"]),
                 ("w_f", w_f), ("stage_rgb", rgb_f), ("stage_t", t_f)):
        dump["%s_%s" % (k, tag)] = v.cpu().numpy()
    print(prec, "fused vs stage chain: rgb %.2e depth %.2e" % (float((out["rgb"] - rgb_f).abs().max()), float((out["t"] - t_f).abs().max())))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "dissect_%s.npz" % scene), **dump)
print("wrote gpurun_out/dissect_%s.npz" % scene)
