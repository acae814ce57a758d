#!/usr/bin/env python3
"""CPU emulation of the precision modes END TO END (no GPU): coarse pass -> weights -> inverse-CDF fine samples -> fine
pass -> composited rgb / depth, against the fp32 oracle, over weight families and seeds.

    python tests/probes/emu_mx_e2e.py [--rays 256] [--seeds 4]

Questions it answers (profiles/r3_precision_emulation.md):
  1. does coarse fp16x3 + fine fp16mx hold 1e-3 on heavy-tailed weights (the headline's parity claim)?
  2. does power-of-two cross-layer equalisation at pack time (mlp_nerf_mx.hip, nerf_mx_pack) restore the margin?
  3. is there a coarse pass cheaper than three fp16 products: the hybrid (Wh.xh + Wl.xh in fp16, Wh.xl in fp6 = 2.25
     MFMA slots), MX-fp8 (e4m3) corrections (2.0 slots), per-layer assignment?
The arithmetic emulated is mlp_mx.h's: fp16 hi/lo split, e2m3 with one E8M0 scale per 32-value block (activations: the
32 values a lane holds; weights: one exponent per output row), products summed in float64 (the fp32 accumulation of
the MFMA is ~1e-7 relative and not what these modes are limited by).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fields, raymarch  # noqa: E402
from tgtc_style_amd import synth  # noqa: E402

ACT_SHIFT = int(os.environ.get("EMU_ACT_SHIFT", "1"))    # activation block scale 2^(E - ACT_SHIFT): 1 = codes in [2,4) (the kernel's), 2 = [4,8)
AL_SHIFT = int(os.environ.get("EMU_AL_SHIFT", "0"))      # lo-activation block scale 2^(E - ACT_SHIFT - 11 - AL_SHIFT): 0 = the round-2 kernel
W_SHIFT = os.environ.get("EMU_W_SHIFT", "1")               # weight row scale: "1", "2", or "best" (per row, least squared error)
E2M3 = np.array([(c & 7) * 0.125 if (c >> 3) == 0 else (1 + (c & 7) / 8) * 2.0 ** ((c >> 3) - 1) for c in range(32)])


def e2m3(x):
    a = np.minimum(np.abs(x), 7.5)
    q = np.where(a < 1, a * 8, np.where(a < 2, 8 + (a - 1) * 8, np.where(a < 4, 16 + (a - 2) * 4, 24 + (a - 4) * 2)))
    return np.sign(x) * E2M3[np.clip(np.rint(q), 0, 31).astype(np.int64)]


def e4m3(x):
    """OCP e4m3fn, round to nearest even, saturating at 448 (3 mantissa bits, per-element exponent)."""
    a = np.minimum(np.abs(x), 448.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -6)))
    q = 2.0 ** (e - 3)
    return np.sign(x) * np.rint(a / q) * q


def act_blocks(K):
    """feature index lists of the (kb, g) blocks of a K-wide activation vector (K = 128 or 256): the 32 values of lane group g"""
    out = []
    for kb in range(K // 128):
        for g in range(4):
            out.append([128 * kb + 32 * s + 16 * h + 4 * g + r for s in range(4) for h in range(2) for r in range(4)])
    return out


def split_act(a, fmt):
    """a [M,K] float32 >= 0 -> Ah, Al (fp16 values), Ah6, Al6 (block-scaled low-precision values), float64"""
    ah = a.astype(np.float16)
    al = (a.astype(np.float32) - ah.astype(np.float32)).astype(np.float16).astype(np.float64)   # may be an fp16 subnormal
    ah = ah.astype(np.float64)
    ah6, al6 = np.zeros_like(ah), np.zeros_like(ah)
    for idx in act_blocks(a.shape[1]):
        m = ah[:, idx].max(1)
        e = np.floor(np.log2(np.maximum(m, 2.0 ** -14)))
        s = (2.0 ** (e - ACT_SHIFT))[:, None]
        if fmt == "e2m3":
            ah6[:, idx] = e2m3(ah[:, idx] / s) * s
            sl = s / 2048 / 2.0 ** AL_SHIFT
            al6[:, idx] = e2m3(al[:, idx] / sl) * sl
        else:   # MX-fp8: block scale puts the block maximum at 2^7 (e4m3 max 448)
            s8 = s / 64
            ah6[:, idx] = e4m3(ah[:, idx] / s8) * s8
            al6[:, idx] = e4m3(al[:, idx] / (s8 / 2048)) * (s8 / 2048)
    return ah, al, ah6, al6


def split_w(W, fmt):
    wh = W.astype(np.float16).astype(np.float64)
    wl = W.astype(np.float64) - wh
    m = np.abs(wh).max(1)
    e = np.floor(np.log2(np.maximum(m, 2.0 ** -14)))
    sh = (2.0 ** (e - 1))[:, None]
    if fmt == "e2m3" and W_SHIFT != "1":
        def q(x, s0):
            cands = [s0, s0 / 2] if W_SHIFT == "best" else [s0 / 2]
            out, err = None, None
            for sc in cands:
                v = e2m3(x / sc) * sc
                er = ((v - x) ** 2).sum(1, keepdims=True)
                out = v if out is None else np.where(er < err, v, out)
                err = er if err is None else np.minimum(er, err)
            return out
        el = np.floor(np.log2(np.maximum(np.abs(wl).max(1), 2.0 ** -40)))
        return wh, wl, q(wh, sh), q(wl, (2.0 ** (el - 1))[:, None])
    if fmt == "e2m3":
        return wh, wl, e2m3(wh / sh) * sh, e2m3(wl / (sh / 2048)) * (sh / 2048)
    s8 = sh / 64
    return wh, wl, e4m3(wh / s8) * s8, e4m3(wl / (s8 / 2048)) * (s8 / 2048)


def make_linear(mode):
    """mode -> f(a, W, b): one dense layer on activations a (float32, >= 0)."""
    if mode == "exact":
        return lambda a, W, b: a.astype(np.float64) @ W.astype(np.float64).T + b
    if mode == "fp16":
        return lambda a, W, b: a.astype(np.float16).astype(np.float64) @ W.astype(np.float16).astype(np.float64).T + b
    if mode == "fp16x3":      # hi*hi + lo*hi + hi*lo with fp16 operands (lo may be an fp16 subnormal, or flush below 6e-8)
        def x3(a, W, b):
            ah = a.astype(np.float16)
            al = (a.astype(np.float32) - ah.astype(np.float32)).astype(np.float16).astype(np.float64)
            wh = W.astype(np.float16)
            wl = (W.astype(np.float32) - wh.astype(np.float32)).astype(np.float16).astype(np.float64)
            ah, wh = ah.astype(np.float64), wh.astype(np.float64)
            return ah @ wh.T + al @ wh.T + ah @ wl.T + b
        return x3
    fmt = "e4m3" if "fp8" in mode else "e2m3"

    def lin(a, W, b):
        ah, al, ah6, al6 = split_act(a, fmt)
        wh, wl, wh6, wl6 = split_w(W, fmt)
        y = ah @ wh.T
        if mode in ("fp16mx", "fp16mx8"):          # main + two low-precision corrections (1.5 / 2.0 MFMA slots)
            y = y + ah6 @ wl6.T + al6 @ wh6.T
        elif mode in ("hybrid", "hybrid8"):        # Wl.xh exact in fp16, Wh.xl low precision (2.25 / 2.5 slots)
            y = y + ah @ wl.astype(np.float16).astype(np.float64).T + al6 @ wh6.T
        elif mode == "hybrid_b":                   # the other way round: Wh.xl in fp16, Wl.xh in fp6 (2.25 slots)
            y = y + ah6 @ wl6.T + al @ wh.T
        else:
            raise ValueError(mode)
        return y + b
    return lin


def equalise(sd, verbose=False):
    """Power-of-two cross-layer equalisation of the ReLU trunk (what nerf_mx_pack does before it packs): row i of hidden
    layer l (and its bias) times 2^-k_i, column i of every consumer of that feature times 2^k_i; exact in floating point,
    the network computes the same function.  k_i balances the row's range against its consumers' column range."""
    sd = {k: v.copy() for k, v in sd.items()}
    W = lambda n: sd["net." + n + ".weight"]

    def consumers(i):
        if i < 7:
            return [("base_layers.%d" % (i + 1), 63 if i == 4 else 0)]
        return [("sigma_layer", 0), ("base_remap_layer", 0)]
    chains = [("base_layers.%d" % i, consumers(i)) for i in range(8)] + [("rgb_layers.0", [("rgb_layers.1", 0)])]
    for name, cons in chains:
        w = W(name)
        n = w.shape[0]
        r1 = np.abs(w).max(1)
        r2 = np.max(np.stack([np.abs(W(c)[:, c0:c0 + n]).max(0) for c, c0 in cons]), 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            k = np.where((r1 > 0) & (r2 > 0), np.rint(0.5 * np.log2(r1 / r2)), 0.0)
        k = k - np.rint(np.median(k))                      # only the spread matters; keep the typical activation scale
        k = np.clip(k, -8, 8)
        s = (2.0 ** -k).astype(np.float32)
        sd["net." + name + ".weight"] = w * s[:, None]
        sd["net." + name + ".bias"] = sd["net." + name + ".bias"] * s
        for c, c0 in cons:
            W(c)[:, c0:c0 + n] *= (1.0 / s)[None, :]
        if verbose:
            print("   equalise %-16s k in [%d, %d]" % (name, k.min(), k.max()))
    return sd


def heavy(sd, seed, family):
    """synth.heavy_tailed; the 'elements' family (a different function) gets its density head re-centred on the base scene"""
    out = synth.heavy_tailed(sd, seed, family)
    return recalibrate_sigma(out, sd) if family == "elements" else out


def recalibrate_sigma(sd, base):
    """Affine correction of the density head so that a changed trunk keeps the base scene's density statistics (mean and
    spread of sigma over the sample points of a few rays): the family changes the numbers the kernels chew, not how well
    conditioned the scene is."""
    ro, rd = rays(48)
    pts, _ = raymarch.sample_coarse(ro, rd, 64, 0.0, 1.0)
    dirs = rd[:, None, :].expand(-1, 64, -1)
    s0 = fields.style_nerf(T(base), pts, dirs)["sigma"]
    s1 = fields.style_nerf(T(sd), pts, dirs)["sigma"]
    g = float(s0.std() / s1.std())
    sd["net.sigma_layer.weight"] = (sd["net.sigma_layer.weight"] * np.float32(g)).astype(np.float32)
    sd["net.sigma_layer.bias"] = (sd["net.sigma_layer.bias"] * np.float32(g) + np.float32(float(s0.mean()) - g * float(s1.mean()))).astype(np.float32)
    return sd


def run_net(sd, pe, de, lin, per_layer=None):
    """MLP_style.forward (models.py:95-117) with `lin` on every activation-input product; PE / direction inputs exact
    (their k-steps keep the three-product fp16 scheme).  per_layer: optional {layer index: lin} overrides."""
    relu = lambda x: np.maximum(x, 0)
    Wm = lambda n: sd["net." + n + ".weight"]
    Bm = lambda n: sd["net." + n + ".bias"].astype(np.float64)
    L = lambda i: (per_layer or {}).get(i, lin)
    h = relu(pe @ Wm("base_layers.0").astype(np.float64).T + Bm("base_layers.0")).astype(np.float32)
    for i in range(7):
        n = "base_layers.%d" % (i + 1)
        w = Wm(n)
        if i == 4:
            y = L(i + 1)(h, w[:, 63:], Bm(n)) + pe @ w[:, :63].astype(np.float64).T
        else:
            y = L(i + 1)(h, w, Bm(n))
        h = relu(y).astype(np.float32)
    sigma = L(8)(h, Wm("sigma_layer"), Bm("sigma_layer"))[:, 0]
    remap = relu(L(9)(h, Wm("base_remap_layer"), Bm("base_remap_layer"))).astype(np.float32)
    w = Wm("rgb_layers.0")
    f = relu(L(10)(remap, w[:, :256], Bm("rgb_layers.0")) + de @ w[:, 256:].astype(np.float64).T).astype(np.float32)
    rgb = 1 / (1 + np.exp(-L(11)(f, Wm("rgb_layers.1"), Bm("rgb_layers.1"))))
    return sigma.astype(np.float32), rgb.astype(np.float32)


def emu_field(sd, pts, dirs, lin, per_layer=None):
    R, N = pts.shape[:2]
    pe = fields.posenc(pts.reshape(-1, 3), 10).to(torch.float32).numpy().astype(np.float64)
    de = fields.posenc(dirs.reshape(-1, 3), 4).to(torch.float32).numpy().astype(np.float64)
    sigma, rgb = run_net(sd, pe, de, lin, per_layer)
    return torch.from_numpy(sigma).reshape(R, N), torch.from_numpy(rgb).reshape(R, N, 3)


def render(sd_c, sd_f, ro, rd, lin_c, lin_f, nc=128, nf=64, per_layer_c=None):
    pts, ts = raymarch.sample_coarse(ro, rd, nc, 0.0, 1.0)
    sig, rgb = emu_field(sd_c, pts, rd[:, None, :].expand(-1, nc, -1), lin_c, per_layer_c)
    _, _, w_c = raymarch.composite(rgb, sig, ts)
    pts_f, ts_f = raymarch.sample_fine(ro, rd, ts, w_c, nf)
    sig, rgb = emu_field(sd_f, pts_f, rd[:, None, :].expand(-1, nc + nf, -1), lin_f)
    rgb_f, t_f, _ = raymarch.composite(rgb, sig, ts_f)
    return rgb_f, t_f


def T(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def rays(n, pose=3, H=400, W=400):
    """n rays spread over a fern-shaped frame (same generator as the oracle's ray test: oracle.rays)"""
    from oracle import rays as orays
    o, d = orays.frame_rays_ndc(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(pose))
    idx = np.linspace(0, H * W - 1, n).astype(np.int64)
    return torch.from_numpy(o[idx]), torch.from_numpy(d[idx])


def stats(rgb, t, ref, alt):
    """max error on the well-conditioned rays (oracle moves < 1e-4 under a 1e-7 shift of the origin), count of the others"""
    e = torch.maximum((rgb - ref["rgb_fine"]).abs().max(-1).values, (t - ref["t_fine"]).abs())
    unstable = torch.zeros_like(e, dtype=torch.bool)
    for a in alt:
        unstable |= torch.maximum((a["rgb_fine"] - ref["rgb_fine"]).abs().max(-1).values, (a["t_fine"] - ref["t_fine"]).abs()) > 1e-4
    return float(e[~unstable].max()), float(e.median()), int(unstable.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=192)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--families", default="base,rows,outliers,elements")
    ap.add_argument("--coarse", default="", help="comma list of coarse-pass modes to try with an exact fine pass (item 3)")
    a = ap.parse_args()
    ro, rd = rays(a.rays)
    exact = make_linear("exact")
    for fam in a.families.split(","):
        for s in range(a.seeds):
            sc, sf = heavy(synth.nerf_state(2 * s), s, fam), heavy(synth.nerf_state(2 * s + 1), s, fam)
            ref = fields.render_plain(T(sc), T(sf), ro, rd, 128, 64)
            alt = [fields.render_plain(T(sc), T(sf), ro * k, rd, 128, 64) for k in (1 + 1e-7, 1 - 1e-7)]
            row = "%-9s seed %d |" % (fam, s)
            for label, net_f, lin_f in (("fine fp16", sf, "fp16"), ("fine mx", sf, "fp16mx"), ("fine mx+equalised", equalise(sf), "fp16mx")):
                rgb, t = render(sc, net_f, ro, rd, exact, make_linear(lin_f))
                mx, med, nu = stats(rgb, t, ref, alt)
                row += " %s: max %.1e med %.1e |" % (label, mx, med)
            for mode in [m for m in a.coarse.split(",") if m]:
                eq = mode.endswith("+eq")
                m0 = mode[:-3] if eq else mode
                rgb, t = render(equalise(sc) if eq else sc, sf, ro, rd, make_linear(m0), exact)
                mx, med, nu = stats(rgb, t, ref, alt)
                row += " coarse %s: max %.1e med %.1e |" % (mode, mx, med)
            print(row + " (%d ill-conditioned rays excluded)" % nu, flush=True)


if __name__ == "__main__":
    main()
