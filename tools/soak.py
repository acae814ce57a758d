#!/usr/bin/env python3
"""Soak test: the hand-rolled LDS ring / barrier protocols must be deterministic.  Renders the same frames over and over
and requires every output to be bit-identical to the first one (a lost chunk, a late barrier or a hazard shows up as a
mismatch on some launch).  tools/soak.py [seconds per case]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import style2d, synth, utils

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
H = W = 400
focal = synth.fern_intrinsics(H, W)
rays = [utils.gen_rays(H, W, focal, synth.spiral_pose(i)) for i in range(3)]


def soak(name, fn, n_inputs):
    ref = [None] * n_inputs
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        i = n % n_inputs
        out = fn(i)
        out = [o.clone() for o in out]
        if ref[i] is None:
            ref[i] = out
        else:
            for a, b in zip(out, ref[i]):
                if not torch.equal(a, b):
                    bad += 1
                    print("  MISMATCH %s launch %d input %d: max |diff| %.3e" % (name, n, i, float((a.float() - b.float()).abs().max())), flush=True)
        n += 1
    torch.cuda.synchronize()
    print("%-34s %5d launches in %5.1f s, %d mismatches" % (name, n, time.time() - t0, bad), flush=True)
    return bad


bad = 0
for prec in ("fp16x3+fp16mx", "fp16x3", "fp16"):
    r = bench.make_renderer(prec, False)
    bad += soak("plain " + prec, lambda i: (lambda o: (o["rgb"], o["t"]))(r.render(*rays[i], 128, 64, near=0., far=1.)), 3)
    if r._split_is_faster():    # the library's choice above was the split path (two-tile kernels); the single ray kernel as well
        from tgtc_style_amd import rendering
        r1 = rendering.RayRenderer(r.coarse, r.fine, fused="single")
        bad += soak("plain " + prec + " (single kernel)", lambda i: (lambda o: (o["rgb"], o["t"]))(r1.render(*rays[i], 128, 64, near=0., far=1.)), 3)
r = bench.make_renderer("fp16x3", True)
z = torch.from_numpy(np.random.default_rng(4).standard_normal((H * W, 32)).astype(np.float32)).cuda()
bad += soak("styled fp16x3", lambda i: (lambda o: (o["rgb"], o["t"]))(r.render(*rays[i], 128, 64, near=0., far=1., z=z)), 3)
os.environ["TGTC_BENCH_CHAIN"] = "1"
r = bench.make_renderer("fp16x3+fp16mx", False)
bad += soak("plain chain fp16x3+fp16mx", lambda i: (lambda o: (o["rgb"], o["t"]))(r.render(*rays[i], 128, 64, near=0., far=1.)), 3)
r = bench.make_renderer("fp16x3", True)
bad += soak("styled chain fp16x3", lambda i: (lambda o: (o["rgb"], o["t"]))(r.render(*rays[i], 128, 64, near=0., far=1., z=z)), 3)
del os.environ["TGTC_BENCH_CHAIN"]

# the fused training kernels: forward outputs and the stashed pre-activation gradients are bit-reproducible; the weight
# gradients are sums of float atomics (order varies) and must agree to rounding
from tgtc_style_amd import fused_train, models
net = models.StyleNerf(bench.NetArgs, mode="fine")
net.load_state_dict(bench.t_state(synth.nerf_state(1)))
net = net.cuda()
tr = fused_train.NerfTrainer()
params = [p.detach().float().contiguous() for p in fused_train.mlp_parameters(net.net)]
M = 50000
gen = torch.Generator(device="cuda").manual_seed(3)
pts = [(torch.rand(M, 3, device="cuda", generator=gen, dtype=torch.float64) * 2.4 - 1.2) for _ in range(2)]
dirs = torch.rand(M, 3, device="cuda", generator=gen, dtype=torch.float64) * 2 - 1
g_rgb = torch.randn(M, 3, device="cuda", generator=gen) * 1e-3
g_sig = torch.randn(M, device="cuda", generator=gen) * 1e-5
grad_ref = {}


def train_once(i):
    ws = tr.lease(pts[i].shape[0], pts[i].device)
    rgb, sigma = tr.forward(params, pts[i], dirs, ws)
    grads = tr.backward(params, rgb, g_rgb, g_sig, ws)
    tr.release(ws)
    tr.status()
    flat = torch.cat([g.reshape(-1) for g in grads])
    if i not in grad_ref:
        grad_ref[i] = flat.clone()
    rel = float((flat - grad_ref[i]).abs().max() / grad_ref[i].abs().max())
    assert rel <= 1e-5, rel
    return rgb, sigma


bad += soak("training forward (+ backward 1e-5)", train_once, 2)

mods = {}
for name, cls, sd in (("tr", style2d.Transformer, synth.transformer_state(5)), ("pe", style2d.PatchEmbed, synth.embed_state(6)),
                      ("dec", style2d.Decoder, synth.decoder_state(7)), ("vgg", style2d.VGG, synth.vgg_state(8))):
    m = cls()
    m.load_state_dict(bench.t_state(sd))
    mods[name] = m.cuda()
net = style2d.StyTrans(mods["vgg"], mods["dec"], mods["pe"], mods["tr"])
contents = [torch.rand(1, 3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(i)) for i in range(3)]
style = torch.from_numpy(synth.style_image(11, H, W)).cuda()
bad += soak("2-D pass", lambda i: style2d.stylize_frame(net, contents[i], style)[:2], 3)
print("soak: %s" % ("OK" if bad == 0 else "%d MISMATCHES" % bad))
sys.exit(1 if bad else 0)
