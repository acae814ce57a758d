"""Reads the dumps of ./mfma_fp6 and checks the layout hypotheses the fp16+fp6 kernels rely on."""
import sys
import numpy as np

d = sys.argv[1] if len(sys.argv) > 1 else "."


def e2m3_decode(code):
    s = np.where(code & 32, -1.0, 1.0)
    e = (code >> 3) & 3
    m = code & 7
    return s * np.where(e == 0, m * 0.125, (1 + m / 8.0) * 2.0 ** (e - 1.0))


def e2m3_encode(x):
    """round to nearest even, saturating"""
    codes = np.arange(32)
    vals = e2m3_decode(codes)
    a = np.minimum(np.abs(x), 7.5)
    idx = np.searchsorted(vals, a)  # vals ascending
    idx = np.clip(idx, 1, 31)
    lo, hi = vals[idx - 1], vals[idx]
    pick_hi = (a - lo > hi - a) | ((a - lo == hi - a) & ((idx & 1) == 0))
    c = np.where(pick_hi, idx, idx - 1)
    return c | np.where(np.signbit(x), 32, 0)


def unpack6(words):  # [64, 6] uint32 -> [64, 32] codes, contiguous little endian bit stream
    out = np.zeros((words.shape[0], 32), dtype=np.int64)
    for l in range(words.shape[0]):
        big = 0
        for j in range(6):
            big |= int(words[l, j]) << (32 * j)
        for i in range(32):
            out[l, i] = (big >> (6 * i)) & 63
    return out


vin = np.fromfile(f"{d}/cvt_in.bin", dtype=np.float16).astype(np.float64).reshape(64, 32)
for s, scale in enumerate([1.0, 4.0, 0.25]):
    w = np.fromfile(f"{d}/cvt_out{s}.bin", dtype=np.uint32).reshape(64, 6)
    codes = unpack6(w)
    for name, x in (("in/scale", vin / scale), ("in*scale", vin * scale)):
        exp = e2m3_encode(x)
        # -0 vs +0 is immaterial
        bad = (e2m3_decode(codes) != e2m3_decode(exp))
        print(f"cvt scale={scale}: hypothesis code = e2m3({name}), contiguous 6-bit LE: {bad.sum()} / {bad.size} mismatches")
        if 0 < bad.sum() < 40:
            for l, i in zip(*np.nonzero(bad)):
                print("   ", l, i, x[l, i], e2m3_decode(codes[l, i]), e2m3_decode(exp[l, i]))

for swap in (0, 1):
    o = np.fromfile(f"{d}/onehot{swap}.bin", dtype=np.float32).reshape(2048, 11, 64, 4)
    # C layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    C = np.zeros((2048, 11, 16, 16))
    for l in range(64):
        for r in range(4):
            C[:, :, 4 * (l >> 4) + r, l & 15] = o[:, :, l, r]
    bad_rows = bad_pair = 0
    examples = []
    for t in range(2048):
        l, i = divmod(t, 32)
        nz = np.nonzero(np.abs(C[t, 0]).sum(axis=(1 - swap)))[0]  # rows (swap=0) or cols (swap=1) hit
        if len(nz) != 1 or nz[0] != l % 16:
            bad_rows += 1
            if len(examples) < 5:
                examples.append(("row", t, nz))
            continue
        line = C[t, :, nz[0], :] if swap == 0 else C[t, :, :, nz[0]]  # [11 bits, 16 others]
        pos = ((line == 2.0).astype(np.int64) << np.arange(11)[:, None]).sum(axis=0)
        k = 32 * (l // 16) + i
        exp = (np.arange(16) + 16 * (k // 32)) * 32 + k % 32
        if not np.array_equal(pos, exp):
            bad_pair += 1
            if len(examples) < 5:
                examples.append(("pair", t, pos[:4], exp[:4]))
    print(f"onehot swap={swap}: one-hot operand lane map wrong for {bad_rows}, pairing wrong for {bad_pair} of 2048", examples)

o = np.fromfile(f"{d}/scale_out.bin", dtype=np.float32).reshape(64, 4)
C = np.zeros((16, 16))
for l in range(64):
    for r in range(4):
        C[4 * (l >> 4) + r, l & 15] = o[l, r]
sa = np.array([127 + (1 if l % 16 == 3 else 0) + (2 if l // 16 == 2 else 0) for l in range(64)])
sb = np.array([127 + (3 if l % 16 == 5 else 0) - (1 if l // 16 == 1 else 0) for l in range(64)])
E = np.zeros((16, 16))
for r in range(16):
    for n in range(16):
        for kb in range(4):
            E[r, n] += 32 * 2.0 ** (sa[r + 16 * kb] - 127) * 2.0 ** (sb[n + 16 * kb] - 127)
print("scale: max |C - expected| =", np.abs(C - E).max(), "(row 3:", C[3, :6], "expected", E[3, :6], ")")
