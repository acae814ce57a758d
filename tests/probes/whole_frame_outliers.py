#!/usr/bin/env python3
"""Which rays of the whole-frame spot check are off, and does the fp32 reference itself move there?
Compares the HIP render with the oracle in float32 AND with the oracle evaluated in float64 (weights and arithmetic):
a ray on which the two oracles disagree by more than the tolerance sits on a discontinuity of the reference
algorithm (last-sample sigma sign, utils.py:367-369; cdf step < 1e-5, utils.py:604-605), not on a kernel error."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fields
from tgtc_style_amd import models, rendering, synth, utils
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_whole_frame_gpu as tw

scene = sys.argv[1] if len(sys.argv) > 1 else "trex"
styled = (sys.argv[2] if len(sys.argv) > 2 else "styled") == "styled"
H, W = tw.SHAPES[scene]
n = H * W
a, (coarse, fine) = tw.nets()
T = tw.T
pose = synth.spiral_pose(17 if styled else 9)
ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), pose)
idx = tw.spot_indices(H, W)
z = None
if styled:
    cm, sm = models.StyleMLP_before_concat(a), models.StyleMLP_Wild_multilayers(a)
    cm.load_state_dict(T(synth.concat_state(2))); sm.load_state_dict(T(synth.style_state(3)))
    lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    lat.load_state_dict(T(synth.latents_state(4))); lat = lat.cuda(); lat.sigma_scale = 1.0
    z = lat(style_ids=torch.zeros(n, dtype=torch.long), frame_ids=torch.full((n,), 17, dtype=torch.long), type="llff")
    r = rendering.RayRenderer(coarse, fine, models.StylePair(cm.cuda(), sm.cuda()))
else:
    r = rendering.RayRenderer(coarse, fine)
out = r.render(ro[idx].contiguous(), rd[idx].contiguous(), 128, 64, z=None if z is None else z[idx].contiguous())

def oracle(dtype):
    c = lambda sd: {k: v.to(dtype) for k, v in T(sd).items()}
    o, d = ro[idx].cpu().to(torch.float64), rd[idx].cpu().to(torch.float64)
    if styled:
        return fields.render_styled(c(synth.nerf_state(0)), c(synth.nerf_state(1)), c(synth.concat_state(2)), c(synth.style_state(3)),
                                    o, d, z[idx].cpu().to(dtype), 128, 64)
    return fields.render_plain(c(synth.nerf_state(0)), c(synth.nerf_state(1)), o, d, 128, 64)

r32 = oracle(torch.float32)
e = (out["rgb"].cpu() - r32["rgb_fine"]).abs().max(-1).values
et = (out["t"].cpu() - r32["t_fine"]).abs()
print("rays %d  rgb err: max %.2e  median %.2e  >1e-3: %d  >1e-4: %d" % (idx.numel(), e.max(), e.median(), int((e > 1e-3).sum()), int((e > 1e-4).sum())))
print("depth err: max %.2e  median %.2e  >1e-3: %d" % (et.max(), et.median(), int((et > 1e-3).sum())))
# sensitivity of the fp32 reference chain itself: the same rays with the origin moved by 1e-7 relative
def oracle_shifted(eps):
    c = lambda sd: T(sd)
    o, d = ro[idx].cpu() * (1.0 + eps), rd[idx].cpu()
    if styled:
        return fields.render_styled(c(synth.nerf_state(0)), c(synth.nerf_state(1)), c(synth.concat_state(2)), c(synth.style_state(3)),
                                    o, d, z[idx].cpu(), 128, 64)
    return fields.render_plain(c(synth.nerf_state(0)), c(synth.nerf_state(1)), o, d, 128, 64)

rs = oracle_shifted(1e-7)
ds = (rs["rgb_fine"] - r32["rgb_fine"]).abs().max(-1).values
dts = (rs["t_fine"] - r32["t_fine"]).abs()
worst = torch.argsort(torch.maximum(e, et), descending=True)[:6]
for k in worst:
    print("ray %6d: hip-vs-oracle rgb %.2e depth %.2e | oracle(o*(1+1e-7)) - oracle(o): rgb %.2e depth %.2e" % (int(idx[k]), e[k], et[k], ds[k], dts[k]))
print("oracle under the 1e-7 shift, all %d rays: rgb max %.2e depth max %.2e, rays moving > 1e-4: %d" % (idx.numel(), ds.max(), dts.max(), int((torch.maximum(ds, dts) > 1e-4).sum())))
