#!/usr/bin/env python3
"""Development probe: where does the fp16 fine pass differ from fp16x3 on a full-frame ray sample?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import fields
from tgtc_style_amd import hip, synth, utils, rendering, models
t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
def net(seed, mode, prec):
    a = type("A", (bench.NetArgs,), {"precision": prec})
    m = models.StyleNerf(a, mode=mode)
    m.load_state_dict(t(synth.nerf_state(seed)))
    return m.cuda()
H = W = 400
ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(3))
idx = torch.arange(0, H * W, H * W // 1024)[:1024].cuda()
ro, rd = ro[idx].contiguous(), rd[idx].contiguous()
outs = {}
for pf in ("fp16x3", "fp16"):
    r = rendering.RayRenderer(net(0, "coarse", "fp16x3"), net(1, "fine", pf))
    outs[pf] = r.render(ro, rd, 128, 64, want_fine_samples=True) if "want_fine_samples" in r.render.__code__.co_varnames else r.render(ro, rd, 128, 64)
a, b = outs["fp16x3"], outs["fp16"]
e = (a["rgb"] - b["rgb"]).abs().max(1).values
print("rays with rgb diff > 1e-2:", int((e > 1e-2).sum()), " > 1e-1:", int((e > 1e-1).sum()), "worst", float(e.max()), "at", int(e.argmax()))
print("keys", list(a.keys()))
w = int(e.argmax())
print("x3 rgb", a["rgb"][w].tolist(), "fp16 rgb", b["rgb"][w].tolist(), "t", float(a["t"][w]), float(b["t"][w]))
# per-sample outputs of the fine net on identical points
lib = hip.load()
ts = torch.sort(torch.rand(1024, 192, device="cuda"), -1)[0]
res = {}
for pf in ("fp16x3", "fp16"):
    n = net(1, "fine", pf)
    rgb = torch.zeros(1024, 192, 3, device="cuda"); sig = torch.zeros(1024, 192, device="cuda")
    hip.check(lib.tgtc_nerf_forward_rays(n.packed().handle, hip.ptr(ro), hip.ptr(rd), hip.ptr(ts), 1024, 192, hip.ptr(rgb), hip.ptr(sig), hip.stream()))
    torch.cuda.synchronize()
    res[pf] = (rgb, sig)
ds = (res["fp16"][1] - res["fp16x3"][1]).abs(); dr = (res["fp16"][0] - res["fp16x3"][0]).abs()
print("per-sample: sigma max diff %.3e (max |sigma| %.3e), rgb max diff %.3e; nonfinite fp16 sigma %d rgb %d" % (float(ds.max()), float(res["fp16x3"][1].abs().max()), float(dr.max()), int((~torch.isfinite(res["fp16"][1])).sum()), int((~torch.isfinite(res["fp16"][0])).sum())))
i = int(ds.argmax()); print("worst sigma sample", i // 192, i % 192, float(res["fp16"][1].flatten()[i]), float(res["fp16x3"][1].flatten()[i]))
