#!/usr/bin/env python3
"""Development timing: ONE launch of the fine pass's per-sample kernel over a whole 400x400 frame (160 000 rays x 192 depths,
tgtc_nerf_forward_rays, fp16mx), i.e. nerf_mx2_kernel (product) or nerf_mx_kernel (-DTGTC_MX2=0 builds), and ONE launch of the
coarse pass's (128 depths, fp16x3, sigma only): nerf_x3s_kernel (product) or nerf_mlp_kernel (-DTGTC_X3S=0 builds)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgtc_style_amd import hip, synth, utils
H = W = 400
N = 192
o, d = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(0))
o, d = o.contiguous(), d.contiguous()
R = o.shape[0]
ts = torch.linspace(0., 1., N, device="cuda").expand(R, N).contiguous()
_, fine = bench.build_nets("fp16x3+fp16mx")
rgb = torch.empty(R * N, 3, device="cuda")
sigma = torch.empty(R * N, device="cuda")
lib = hip.load()
call = lambda: hip.check(lib.tgtc_nerf_forward_rays(fine.packed().handle, hip.ptr(o), hip.ptr(d), hip.ptr(ts), R, N, hip.ptr(rgb), hip.ptr(sigma), hip.stream()))
for _ in range(2):
    call()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(4):
    call()
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 4
flop = 2.0 * bench.MAC_FULL * R * N
print("fine pass kernel %7.2f ms   %.1f TFLOP/s algorithmic   matrix pipe %.3f   checksum %.6e %.6e" % (
    ms, flop / ms / 1e9, flop * 1.5 / (ms * 1e-3) / 2.5e15, float(rgb.double().sum()), float(sigma.double().sum())), flush=True)

NC = 128
coarse, _ = bench.build_nets("fp16x3+fp16mx")
ts_c = torch.linspace(0., 1., NC, device="cuda").expand(R, NC).contiguous()
sig_c = torch.empty(R * NC, device="cuda")
call_c = lambda: hip.check(lib.tgtc_nerf_forward_rays(coarse.packed().handle, hip.ptr(o), hip.ptr(d), hip.ptr(ts_c), R, NC, None, hip.ptr(sig_c), hip.stream()))
for _ in range(2):
    call_c()
torch.cuda.synchronize()
ev[0].record()
for _ in range(4):
    call_c()
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 4
flop = 2.0 * bench.MAC_SIGMA * R * NC
print("coarse pass kernel %6.2f ms   %.1f TFLOP/s algorithmic   matrix pipe %.3f   checksum %.6e" % (
    ms, flop / ms / 1e9, flop * 3.0 / (ms * 1e-3) / 2.5e15, float(sig_c.double().sum())), flush=True)
