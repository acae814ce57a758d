#!/usr/bin/env python3
"""CPU emulation of the fp16+fp6 arithmetic (mlp_mx.h) on the synthetic NeRF: what error does the scheme itself leave?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fields
from tgtc_style_amd import synth

VALS = np.array([(c & 7) * 0.125 if (c >> 3) == 0 else (1 + (c & 7) / 8) * 2.0 ** ((c >> 3) - 1) for c in range(32)])

def e2m3(x):
    a = np.minimum(np.abs(x), 7.5)
    q = np.where(a < 1, a * 8, np.where(a < 2, 8 + (a - 1) * 8, np.where(a < 4, 16 + (a - 2) * 4, 24 + (a - 4) * 2)))
    c = np.clip(np.rint(q), 0, 31).astype(np.int64)
    return np.sign(x) * VALS[c]

def blocks():
    """feature index lists of the 8 (kb, g) blocks of a 256-wide activation vector"""
    out = []
    for kb in range(2):
        for g in range(4):
            out.append([128 * kb + 32 * s + 16 * h + 4 * g + r for s in range(4) for h in range(2) for r in range(4)])
    return out

def split_act(a, variant):
    """a: [M, K] float32 non-negative -> (Ah, Ah6, Al6) as float64 values"""
    ah = a.astype(np.float16).astype(np.float64)
    lo = ((a.astype(np.float32) - ah.astype(np.float32)) * np.float32(2048)).astype(np.float16).astype(np.float64)
    ah6 = np.zeros_like(ah); al6 = np.zeros_like(ah)
    K = a.shape[1]
    for idx in blocks():
        idx = [i for i in idx if i < K]
        if not idx: continue
        m = ah[:, idx].max(1)
        e = np.floor(np.log2(np.maximum(m, 2.0 ** -14)))
        s = 2.0 ** (e - 1)
        ah6[:, idx] = e2m3(ah[:, idx] / s[:, None]) * s[:, None]
        al6[:, idx] = e2m3(lo[:, idx] / s[:, None]) * s[:, None] / 2048
    return ah, ah6, al6

def split_w(W):
    wh = W.astype(np.float16).astype(np.float64)
    wl = (W.astype(np.float32) - wh.astype(np.float32)).astype(np.float64)
    m = np.abs(wh).max(1)
    e = np.floor(np.log2(np.maximum(m, 2.0 ** -14)))
    sh = 2.0 ** (e - 1)
    sl = sh / 2048
    wh6 = e2m3(wh / sh[:, None]) * sh[:, None]
    wl6 = e2m3(wl / sl[:, None]) * sl[:, None]
    return wh, wl6, wh6

def mx_linear(a, W, b, terms=(1, 1, 1)):
    ah, ah6, al6 = split_act(a, None)
    wh, wl6, wh6 = split_w(W)
    y = ah @ wh.T
    if terms[1]: y = y + ah6 @ wl6.T
    if terms[2]: y = y + al6 @ wh6.T
    return y + b

def run(sd, pe, de, lin):
    relu = lambda x: np.maximum(x, 0)
    W = lambda n: sd["net." + n + ".weight"].astype(np.float64); B = lambda n: sd["net." + n + ".bias"].astype(np.float64)
    h = relu(pe @ W("base_layers.0").T + B("base_layers.0")).astype(np.float32)
    for i in range(7):
        n = "base_layers.%d" % (i + 1)
        if i == 4:
            w = W(n)
            y = lin(h, w[:, 63:].astype(np.float32), B(n)) + pe @ w[:, :63].T
        else:
            y = lin(h, W(n).astype(np.float32), B(n))
        h = relu(y).astype(np.float32)
    sigma = lin(h, W("sigma_layer").astype(np.float32), B("sigma_layer"))[:, 0]
    remap = relu(lin(h, W("base_remap_layer").astype(np.float32), B("base_remap_layer"))).astype(np.float32)
    w = W("rgb_layers.0")
    f = relu(lin(remap, w[:, :256].astype(np.float32), B("rgb_layers.0")) + de @ w[:, 256:].T).astype(np.float32)
    rgb = 1 / (1 + np.exp(-lin(f, W("rgb_layers.1").astype(np.float32), B("rgb_layers.1"))))
    return sigma, remap, rgb

rng = np.random.default_rng(0)
M = 1000
pts = rng.uniform(-1, 1, (M, 3)); dirs = rng.uniform(-1, 1, (M, 3))
sd = synth.nerf_state(1)
pe = fields.posenc(torch.from_numpy(pts), 10).numpy().astype(np.float32).astype(np.float64)
de = fields.posenc(torch.from_numpy(dirs), 4).numpy().astype(np.float32).astype(np.float64)
exact = lambda a, W, b: a.astype(np.float64) @ W.astype(np.float64).T + b
ref = run(sd, pe, de, exact)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for name, lin in (("fp16+fp6 (all terms)", mx_linear), ("no corr1 (Wl.Ah)", lambda a, W, b: mx_linear(a, W, b, (1, 0, 1))),
                  ("no corr2 (Wh.Al)", lambda a, W, b: mx_linear(a, W, b, (1, 1, 0))), ("fp16 only", lambda a, W, b: mx_linear(a, W, b, (1, 0, 0)))):
    out = run(sd, pe, de, lin)
    print("%-24s sigma %.2e  remap %.2e  rgb %.2e" % (name, rel(out[0], ref[0]), rel(out[1], ref[1]), rel(out[2], ref[2])))
