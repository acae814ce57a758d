#!/usr/bin/env python3
"""Generator of the fp16mx (TGTC_PREC_FP16_FP6) NeRF pass on TWO column tiles per wave, one wave per SIMD.

    tools/gen_mx2_asm.py > tgtc-style_amd/csrc/mx2_asm_nerf.inc          (per-sample kernel csrc/mlp_nerf_mx2.hip)

Why.  At one column tile per wave the fp16mx loop is bound by LDS bandwidth (profiles/r4_kernel_variants.md section 1: eight
waves read the same 7 KiB of weight fragments per six MFMAs, 67 % LDS array busy at 0.40 of the matrix pipe).  LDS bytes per
MFMA halve with two column tiles per wave, but the registers (339) only exist with ONE wave per SIMD: 256 architectural VGPRs
plus 256 accumulator registers.  No compiler would place them, so the whole pass is one stream again (tools/gen_mx_asm.py
PassGen is the one-tile version; same arithmetic, same order per sample, bit-identical results):

  VGPR  v16..v23   sigma[ct], rgb[ct][3] (outputs)          AGPR  a0..a55     two weight-group buffers (ds_read straight into
        v24..v31   ReLU'd accumulator values, 4 per tile                      them; A operand of every MFMA)
        v32..v34   row-exponent words of the three acc sets       a56..a151   the e2m3 copies of the activations (B operand of
        v35..v36   running block maximum per tile                             the fp6 MFMAs): [set][tile][h6 12 | l6 12]
        v37..v44   scratch of the block close, 4 per tile          a152..a199  the encodings' fp16 hi / lo B fragments per tile:
        v45..v52   activation block scales [set][tile][block]                 pe hi 8 | pe lo 8 | dir hi 4 | dir lo 4
        v54..v65   e2m3 conversion results on their way to AGPRs
        v66..v89   three accumulator sets x two tiles
        v90..v121  lo-half staging of the open block, 16 per tile
        v122..v249 fp16 hi activations [set][tile][32]  (at entry: the encodings, copied to a152.. first)
  (register tuples are even-aligned: gfx950 requires it of every 64-bit-or-wider operand)

Measured basis (tools/microbench/agpr_mfma.hip): MFMAs whose A operand and fp6 B operand are AGPRs run no slower than from
VGPRs in this pattern (0.85 of the matrix pipe against 0.81 for two alternating chains, one wave per SIMD).

Ring protocol: as tools/gen_mx_asm.py, with four waves (four LDS-DMA instructions per wave and 16 KiB chunk) and the WHOLE
persistent protocol inside the stream: entering the pass (virtual chunk 0), every boundary, and the walk to the end of the
padded stream; the look-ahead runs into the same stream again (the kernel is a loop over passes of one network).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_mx_asm import CHUNK, KGROUP, LOOK, NERF_SHAPES, RING, SLOTS, Emitter, Table  # noqa: E402

NWAVES = 4
GPC = CHUNK // (NWAVES * 1024)
NCT = 2

# ---------------------------------------------------------------------------------------------- registers
OUT = 16
T = 24
RS = 32
MXK = 35
MS = 37
SC = 45
TMP6 = 54
ACC = 66
L16 = 90
H = 122
AW = (0, 28)
A6 = 56
AKEEP = 152
N_AGPR = 200
LAST_VGPR = 249


def V(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def A(base, n=1):
    return "a%d" % base if n == 1 else "a[%d:%d]" % (base, base + n - 1)


def acc(s, ct):
    return ACC + (s * NCT + ct) * 4


def hreg(st, ct):
    return H + (st * NCT + ct) * 32


def a6(st, ct):
    return A6 + (st * NCT + ct) * 24


def sc(st, ct, kb):
    return SC + st * 4 + ct * 2 + kb


KEEP_OFF = {"pe": (0, 8), "dir": (16, 20)}     # (hi, lo) offsets inside a tile's 24 keep registers


class Layer:
    def __init__(self, name, rt, nkb, npe, src, sink, pe):
        self.name, self.rt, self.nkb, self.npe, self.src, self.sink, self.pe = name, rt, nkb, npe, src, sink, pe


def nerf_full_layers():
    return [Layer("L0", 16, 0, 2, None, 0, "pe"), Layer("L1", 16, 2, 0, 0, 1, None), Layer("L2", 16, 2, 0, 1, 0, None),
            Layer("L3", 16, 2, 0, 0, 1, None), Layer("L4", 16, 2, 0, 1, 0, None), Layer("L5", 16, 2, 2, 0, 1, "pe"),
            Layer("L6", 16, 2, 0, 1, 0, None), Layer("L7", 16, 2, 0, 0, 1, None), Layer("SIG", 1, 2, 0, 1, "sigma", None),
            Layer("REMAP", 16, 2, 0, 1, 0, None), Layer("C0", 8, 2, 1, 0, 1, "dir"), Layer("C1", 1, 1, 0, 1, "rgb", None)]


class Pass2:
    def __init__(self, table, layers, nq, padc, cfg=None, nkeep=24):
        self.t, self.layers, self.nq, self.padc = table, layers, nq, padc
        self.nkeep = nkeep           # encoding registers per tile handed over: 24 (point + direction) or 16 (sigma-only pass)
        self.cfg = cfg or {}         # timing experiments (results wrong by construction): abl_epi, abl_reads, abl_ring, abl_barrier, abl_dma, abl_cvt
        self.e = Emitter()
        self.unit_op, self.unit_chunk = {}, {}
        self.dma_q = []
        self.dma_n = self.dma_tick = 0
        self.entered = -1
        self.exit_entered = max(table.chunk_hi(nq - 1) - 1, 0)

    # ------------------------------------------------------------------------------------------ ring
    def boundary(self, v):
        e = self.e
        self.drain_dma(99)                   # (m0 and s[96:97] are about to change)
        victims = [op for op, ch in self.unit_chunk.items() if ch <= v - 1 and op >= e.lds_done]
        if victims:
            e.need(max(victims))
        if self.cfg.get("abl_ring"):
            self.entered = v
            return
        e.emit("s_waitcnt vmcnt(%d)" % ((LOOK - 2) * GPC))
        if not self.cfg.get("abl_barrier"):
            e.emit("s_barrier")
        if self.cfg.get("abl_dma"):
            self.entered = v
            return
        ch = (v + LOOK) % self.padc          # beyond the padded end: the same stream again (the next pass)
        slot = (v + LOOK) % SLOTS
        assert self.padc % SLOTS == 0
        e.emit("s_add_u32 s96, %%[src_lo], 0x%x" % (ch * CHUNK))
        e.emit("s_addc_u32 s97, %[src_hi], 0")
        e.emit("s_add_u32 m0, %%[ldsw], 0x%x" % (slot * CHUNK))
        e.emit("s_nop 0")
        loads = []
        for j in range(GPC):
            if j == 4:                       # the instruction's offset field ends at 4095: second half of a 32 KiB chunk's piece
                loads += ["s_add_u32 s96, s96, 0x1000", "s_addc_u32 s97, s97, 0", "s_add_u32 m0, m0, 0x1000", "s_nop 0"]
            loads.append("global_load_lds_dwordx4 %%[voff], s[96:97] offset:%d" % ((j % 4) * 1024))
        if self.cfg.get("stagger_dma", 0) and GPC == 4:
            # The four waves run this stream in lock step, so "one LDS-DMA per MFMA gap" is four waves at the texture path at once.
            # Staggered: wave w issues its four pieces behind the (4w)th gap after the boundary and skips the other three
            # places with a scalar branch -- at any moment one wave is issuing.
            self.dma_n += 1
            for w in range(NWAVES):
                lab = ".Ldma_%d_%d_%%=" % (self.dma_n, w)
                self.dma_q.append(["s_cmp_eq_u32 %%[wave], %d" % w, "s_cbranch_scc0 %s" % lab] + loads + [lab + ":"])
        elif self.cfg.get("spread_dma", 1):
            unit = []
            for ln in loads:                 # (the scalar set-up lines in front of a load go with it)
                unit.append(ln)
                if ln.startswith("global_load"):
                    self.dma_q.append(unit)
                    unit = []
        else:
            for ln in loads:
                e.emit(ln)
        self.entered = v

    def drain_dma(self, n=1):
        """issue up to n queued LDS-DMA units (staggered: every `stagger_gap`-th call only)"""
        if n == 1 and self.cfg.get("stagger_dma", 0):
            self.dma_tick += 1
            if self.dma_tick % self.cfg.get("stagger_gap", 4):
                return
        while self.dma_q and n > 0:
            for line in self.dma_q.pop(0):
                self.e.emit(line)
            n -= 1

    def acquire_for(self, q):
        g = min(q, self.nq - 1)
        want = self.t.chunk_hi(g) - 1
        while self.entered < want:
            self.boundary(self.entered + 1)

    # ------------------------------------------------------------------------------------------ weight reads
    def wbase(self, q):
        return AW[q & 1]

    def units(self, q):
        n = self.t.npe[q]
        if n:
            return [(j, 4 * j, 4, 1024 * j, True) for j in range(2 * n)]
        return [(0, 0, 4, 0, True), (1, 4, 4, 1024, True), (2, 8, 4, 2048, True), (3, 12, 4, 3072, True),
                (4, 16, 4, 4096, True), (6, 20, 4, 5120, True), (5, 24, 4, 6144, True)]      # fp6 pieces A, B, C (mlp_mx.h)

    def read_unit_at(self, q, unit):
        u, roff, nreg, boff, wide = unit
        t, e = self.t, self.e
        assert t.chunk_hi(q) - 1 <= self.entered, "group %d read before its chunks were entered" % q
        if self.cfg.get("abl_reads") and q >= 2:
            self.unit_op[(q, u)] = None
            return
        o = t.off[q] % RING
        w = self.wbase(q)
        off = o + boff
        if wide:
            base = "%[lane_lo]" if off < 65536 else "%[lane_hi]"
            op = e.lds("ds_read_b128 %s, %s offset:%d" % (A(w + roff, 4), base, off % 65536))
        else:
            base = "%[b8_lo]" if off < 65536 else "%[b8_hi]"
            op = e.lds("ds_read_b64 %s, %s offset:%d" % (A(w + roff, 2), base, off % 65536))
        self.unit_op[(q, u)] = op
        self.unit_chunk[op] = (t.off[q] + boff) // CHUNK

    # ------------------------------------------------------------------------------------------ MFMAs of one group
    def mfmas(self, q, layer, s):
        """the MFMA instructions of group q (both column tiles) with, for each, the buffer registers it is the LAST reader of"""
        w = self.wbase(q)
        n = self.t.npe[q]
        out = []
        f16 = "v_mfma_f32_16x16x32_f16 %s, %s, %s, %s"
        if n:
            hi0, lo0 = KEEP_OFF[layer.pe]
            for k in range(n):
                uh, ul = w + 8 * k, w + 8 * k + 4
                for which in range(3):
                    for ct in range(NCT):
                        a = V(acc(s, ct), 4)
                        ph, pl = A(AKEEP + 24 * ct + hi0 + 4 * k, 4), A(AKEEP + 24 * ct + lo0 + 4 * k, 4)
                        last = ct == NCT - 1
                        if which == 0:
                            out.append((f16 % (a, A(uh, 4), ph, a), []))
                        elif which == 1:
                            out.append((f16 % (a, A(ul, 4), ph, a), list(range(8 * k + 4, 8 * k + 8)) if last else []))
                        else:
                            out.append((f16 % (a, A(uh, 4), pl, a), list(range(8 * k, 8 * k + 4)) if last else []))
            return out
        kb = self.group_kb[q]
        x = layer.src
        sel = "op_sel:[%d,%d,0] op_sel_hi:[0,0,0] cbsz:2 blgp:2"
        fp6 = "v_mfma_scale_f32_16x16x128_f8f6f4 %s, %s, %s, %s, %s, %s " + sel
        for kind, idx in (("M", 0), ("C", 1), ("M", 1), ("M", 2), ("C", 2), ("M", 3)):
            for ct in range(NCT):
                a = V(acc(s, ct), 4)
                last = ct == NCT - 1
                if kind == "M":
                    out.append((f16 % (a, A(w + 4 * idx, 4), V(hreg(x, ct) + 4 * (4 * kb + idx), 4), a),
                                list(range(4 * idx, 4 * idx + 4)) if last else []))
                elif idx == 1:
                    out.append((fp6 % (a, A(w + 16, 6), A(a6(x, ct) + 6 * kb, 6), a, V(RS + s), V(sc(x, ct, kb)), 1, 0),
                                list(range(16, 22)) if last else []))
                else:
                    out.append((fp6 % (a, A(w + 22, 6), A(a6(x, ct) + 12 + 6 * kb, 6), a, V(RS + s), V(sc(x, ct, kb)), 0, 1),
                                list(range(22, 28)) if last else []))
        return out

    # ------------------------------------------------------------------------------------------ epilogue (mx_store_act)
    def epi_ops(self, rt, st, s):
        """row tile rt's epilogue into activation set st from accumulator set s: fillers of both tiles, "ACC_FREE" once"""
        e = self.e
        ks = rt // 2
        head, body = [], [[], []]
        for ct in range(NCT):
            a, t = acc(s, ct), T + 4 * ct
            for i in range(4):
                head.append(lambda i=i, a=a, t=t: e.emit("v_max_i32_e32 %s, 0, %s" % (V(t + i), V(a + i))))
            ops = body[ct]
            stages = [[], [], [], []]        # per half: convert, lo half 0, running maximum, lo half 1 -- each reads the one before
            for half in range(2):
                d = (rt & 1) * 2 + half
                h = hreg(st, ct) + 4 * ks + d
                lo = L16 + 16 * ct + 4 * (ks & 3) + d
                t0, t1 = t + 2 * half, t + 2 * half + 1
                mxk = MXK + ct
                stages[0].append(lambda h=h, t0=t0, t1=t1: e.emit("v_cvt_pk_f16_f32 %s, %s, %s" % (V(h), V(t0), V(t1))))
                stages[1].append(lambda lo=lo, t0=t0, h=h: e.emit(
                    "v_fma_mixlo_f16 %s, %s, 1.0, -%s op_sel:[0,0,0] op_sel_hi:[0,0,1]" % (V(lo), V(t0), V(h))))
                if (rt & 7) == 0 and half == 0:
                    stages[2].append(lambda h=h, mxk=mxk: e.emit("v_mov_b32_e32 %s, %s" % (V(mxk), V(h))))
                else:
                    stages[2].append(lambda h=h, mxk=mxk: e.emit("v_pk_max_u16 %s, %s, %s" % (V(mxk), V(mxk), V(h))))
                stages[3].append(lambda lo=lo, t1=t1, h=h: e.emit(
                    "v_fma_mixhi_f16 %s, %s, 1.0, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (V(lo), V(t1), V(h))))
            if self.cfg.get("reorder_epi"):  # both halves stage by stage (measured: no difference, profiles/r4_kernel_variants.md)
                for stage in stages:
                    ops.extend(stage)
            else:
                for half in range(2):
                    ops.extend(stage[half] for stage in stages)
            if (rt & 7) == 7:
                kb = rt // 8
                m0, m1, m2, m3 = MS + 4 * ct, MS + 4 * ct + 1, MS + 4 * ct + 2, MS + 4 * ct + 3
                mxk, tmp = MXK + ct, TMP6 + 6 * ct
                ops.append(lambda m0=m0, mxk=mxk: e.emit(
                    "v_max_u32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" % (V(m0), V(mxk), V(mxk))))
                ops.append(lambda m0=m0: e.emit("v_lshrrev_b32_e32 %s, 10, %s" % (V(m0), V(m0))))
                ops.append(lambda m0=m0, m1=m1: e.emit("v_add_u32_e32 %s, 0x63, %s" % (V(m1), V(m0))))      # byte_l = byte_h - 12
                ops.append(lambda m1=m1, m3=m3: e.emit("v_lshlrev_b32_e32 %s, 23, %s" % (V(m3), V(m1))))
                ops.append(lambda ct=ct, tmp=tmp, m3=m3: e.emit("v_cvt_scalef32_pk32_fp6_f16 %s, %s, %s" % (V(tmp, 6), V(L16 + 16 * ct, 16), V(m3))))
                ops.append(lambda m0=m0: e.emit("v_add_u32_e32 %s, 0x6f, %s" % (V(m0), V(m0))))              # byte_h
                ops.append(lambda m0=m0, m2=m2: e.emit("v_lshlrev_b32_e32 %s, 23, %s" % (V(m2), V(m0))))
                ops.append(lambda ct=ct, kb=kb, m0=m0, m1=m1: e.emit("v_lshl_or_b32 %s, %s, 8, %s" % (V(sc(st, ct, kb)), V(m1), V(m0))))
                for i in range(6):
                    ops.append(lambda i=i, ct=ct, kb=kb, tmp=tmp: e.emit("v_accvgpr_write_b32 %s, %s" % (A(a6(st, ct) + 12 + 6 * kb + i), V(tmp + i))))
                ops.append(lambda ct=ct, kb=kb, tmp=tmp, m2=m2: e.emit("v_cvt_scalef32_pk32_fp6_f16 %s, %s, %s" % (
                    V(tmp, 6), V(hreg(st, ct) + 16 * kb, 16), V(m2))))
                for i in range(6):
                    ops.append(lambda i=i, ct=ct, kb=kb, tmp=tmp: e.emit("v_accvgpr_write_b32 %s, %s" % (A(a6(st, ct) + 6 * kb + i), V(tmp + i))))
        merged = []
        for i in range(max(len(body[0]), len(body[1]))):
            for ct in range(NCT):
                if i < len(body[ct]):
                    merged.append(body[ct][i])
        return head + ["ACC_FREE"] + merged

    # ------------------------------------------------------------------------------------------ the pass
    def generate(self):
        e, t = self.e, self.t
        dist = 2
        self.group_kb = {}
        rows = []
        q = 0
        bias0 = 0
        for li, L in enumerate(self.layers):
            L.bias0 = bias0
            bias0 += 16 * L.rt
            for rt in range(L.rt):
                gs = []
                for kb in range(L.nkb):
                    assert t.npe[q] == 0
                    self.group_kb[q] = kb
                    gs.append(q)
                    q += 1
                if L.npe:
                    assert t.npe[q] == L.npe, (L.name, q, t.npe[q])
                    gs.append(q)
                    q += 1
                rows.append((li, rt, gs))
        assert q == self.nq, (q, self.nq)
        total_rt = len(rows)
        bias_op = {}

        def load_bias(grt):
            if grt >= total_rt:
                return
            li, rt, _ = rows[grt]
            L = self.layers[li]
            boff = L.bias0 + 16 * rt
            for ct in range(NCT):
                bias_op[grt] = e.lds("ds_read_b128 %s, %%[bias_lane] offset:%d" % (V(acc(grt % 3, ct), 4), boff * 4))
            if L.nkb:
                bias_op[grt] = e.lds("ds_read_u16 %s, %%[rs_lane] offset:%d" % (V(RS + grt % 3), boff * 2))

        fillers = []
        tail_of = [None]

        def run_filler():
            f = fillers.pop(0)
            if isinstance(f, tuple) and f[0] == "ACC_FREE":
                load_bias(f[1] + 3)
            elif self.cfg.get("abl_epi"):
                pass
            else:
                n0 = len(e.lines)
                f()
                if self.cfg.get("abl_cvt") and "v_cvt_scalef32" in e.lines[-1]:
                    del e.lines[n0:]

        def flush(gap=99):
            if fillers and gap < 4:      # the fillers start by reading an accumulator: drain the matrix pipe if it was written just now
                e.emit("s_nop 7")
                e.emit("s_nop 7")
            while fillers:
                run_filler()
            tail_of[0] = None

        def epilogue(grt):
            li, rt, _ = rows[grt]
            L = self.layers[li]
            s = grt % 3
            if L.sink == "sigma":
                return [lambda ct=ct: e.emit("v_mov_b32_e32 %s, %s" % (V(OUT + ct), V(acc(s, ct)))) for ct in range(NCT)] + [("ACC_FREE", grt)]
            if L.sink == "rgb":
                return [lambda ct=ct, i=i: e.emit("v_mov_b32_e32 %s, %s" % (V(OUT + 2 + 3 * ct + i), V(acc(s, ct) + i)))
                        for ct in range(NCT) for i in range(3)] + [("ACC_FREE", grt)]
            return [("ACC_FREE", grt) if f == "ACC_FREE" else f for f in self.epi_ops(rt, L.sink, s)]

        # ---- prologue: the encodings to their AGPRs, enter the pass, biases of three row tiles, groups 0 and 1
        for ct in range(NCT):
            for i in range(self.nkeep):
                e.emit("v_accvgpr_write_b32 %s, %s" % (A(AKEEP + 24 * ct + i), V(H + 24 * ct + i)))
        self.boundary(0)
        for r in range(3):
            load_bias(r)
        for g0 in (0, 1):
            self.acquire_for(g0)
            for unit in self.units(g0):
                self.read_unit_at(g0, unit)

        free = {0: set(), 1: set()}
        for grt, (li, rt, gs) in enumerate(rows):
            L = self.layers[li]
            s = grt % 3
            if grt > 0:
                pli, prt, _ = rows[grt - 1]
                PL = self.layers[pli]
                fillers.extend(epilogue(grt - 1))
                if isinstance(PL.sink, int):
                    tail_of[0] = (PL.sink, prt // 8)
            per = lambda g: NCT * (3 * t.npe[g] if t.npe[g] else 6)     # noqa: E731
            n_row = sum(per(g) for g in gs)
            gap = 0
            for g in gs:
                if t.npe[g] == 0 and tail_of[0] == (L.src, self.group_kb[g]):
                    flush(gap)
                if t.npe[g] == 0 and fillers and tail_of[0] is not None and tail_of[0][0] == L.src and L.nkb == 1:
                    flush(gap)
                self.acquire_for(g + dist)
                ops_needed = [self.unit_op[(g, u[0])] for u in self.units(g) if self.unit_op[(g, u[0])] is not None]
                if g == gs[0]:
                    ops_needed.append(bias_op[grt])
                if ops_needed:
                    e.need(max(ops_needed))
                par = g & 1
                used = set()
                for u in self.units(g):
                    used |= set(range(u[1], u[1] + u[2]))
                free[par] = set(range(28)) - used
                nxt = g + dist
                pending = list(self.units(nxt)) if nxt < self.nq else []

                def issue_ready():
                    for unit in list(pending):
                        if set(range(unit[1], unit[1] + unit[2])) <= free[par]:
                            self.read_unit_at(nxt, unit)
                            pending.remove(unit)
                issue_ready()
                for text, released in self.mfmas(g, L, s):
                    e.emit(text)
                    self.drain_dma()
                    free[par] |= set(released)
                    issue_ready()
                    if gap >= 3 and fillers:
                        remaining = max(n_row - 1 - gap, 0) + 1
                        if tail_of[0] is not None and tail_of[0][0] == L.src:
                            first_dep, seen = 0, 0
                            for gg in gs:
                                if t.npe[gg] == 0 and self.group_kb[gg] == tail_of[0][1]:
                                    first_dep = seen
                                    break
                                seen += per(gg)
                            remaining = max(first_dep - 1 - gap, 0) + 1
                        n = (len(fillers) + remaining - 1) // remaining
                        for _ in range(n):
                            if fillers:
                                run_filler()
                    gap += 1
                assert not pending, "units of group %d could not be placed" % nxt
            if not fillers:
                tail_of[0] = None
        e.emit("s_nop 7")
        e.emit("s_nop 7")
        fillers.extend(epilogue(total_rt - 1))
        flush()
        # walk to the end of the padded stream: the state the next pass's entry expects
        e.emit("s_waitcnt lgkmcnt(0)")
        e.lds_done = e.lds_issued
        for v in range(self.entered + 1, self.padc):
            self.boundary(v)
        self.drain_dma(99)
        return e.lines


def block_text(lines):
    return "\n".join('        "%s\\n\\t"' % ln for ln in lines)


def emit_pass(name, table, layers, nq, units, out, cfg=None, full=True):
    nchunk = (units * 1024 + CHUNK - 1) // CHUNK
    padc = (nchunk + SLOTS - 1) // SLOTS * SLOTS
    nfrag = 6 if full else 4                     # keep fragments per tile: Ph0 Ph1 Pl0 Pl1 [Dh Dl]
    gen = Pass2(table, layers, nq, padc, cfg, nkeep=4 * nfrag)
    lines = gen.generate()
    stats = {}
    for ln in lines:
        k = ln.split()[0]
        stats[k] = stats.get(k, 0) + 1
    out.append("// pass %s: %d groups, %d layers, %d chunks (padded %d); %d instructions; %s" % (
        name, nq, len(layers), nchunk, padc, len(lines), ", ".join("%s %d" % kv for kv in sorted(stats.items(), key=lambda kv: -kv[1])[:14])))
    if full:
        out.append("// keep[ct][0..5] = Ph[0], Ph[1], Pl[0], Pl[1], Dh, Dl of column tile ct (fp16 hi / lo B fragments of the encodings), handed over in")
        out.append("// v122..v169 and moved to a152..a199 by the stream; sigma[ct] (lanes 0..15) and rows 0..2 of the colour head's accumulator come")
        out.append("// back in v16..v23.  The stream runs the whole ring protocol of one pass (entry, boundaries, walk to the padded end).")
        out.append("constexpr int kMx2PadChunks = %d, kMx2ChunkBytes = %d;" % (padc, CHUNK))
    else:
        out.append("// the trunk + sigma head alone (the first nine layers of the same stream): keep[ct][0..3] = Ph[0], Ph[1], Pl[0], Pl[1]; sigma[ct] in v16, v17")
        out.append("constexpr int kMx2SigmaPadChunks = %d;" % padc)
    out.append("template <class Reader>")
    tail = ", float (&rgb)[2][3]" if full else ""
    out.append("__device__ __forceinline__ void mx2_asm_%s(const Reader& rd, int wave, lds_cptr bias_lane, lds_cptr rs_lane, half8 (&keep)[2][%d], float (&sigma)[2]%s) {" % (name, nfrag, tail))
    out.append("    const unsigned src_lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)rd.ring.src[0]), src_hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)rd.ring.src[0] >> 32));")
    out.append("    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(size_t)TGTC_LPTR(rd.ring.lds_wave));")

    def rg(base, n):
        return "{v[%d:%d]}" % (base, base + n - 1)
    outs = ['"=&{v%d}"(sigma[%d])' % (OUT + ct, ct) for ct in range(NCT)]
    if full:
        outs += ['"=&{v%d}"(rgb[%d][%d])' % (OUT + 2 + 3 * ct + i, ct, i) for ct in range(NCT) for i in range(3)]
    outs += ['"+%s"(keep[%d][%d])' % (rg(H + 24 * ct + 4 * i, 4), ct, i) for ct in range(NCT) for i in range(nfrag)]
    out.append("    asm volatile(")
    out.append(block_text(lines))
    ins = ['[lane_lo] "v"(rd.ring.lane_lo)', '[lane_hi] "v"(rd.ring.lane_hi)', '[b8_lo] "v"(rd.b8_lo)', '[b8_hi] "v"(rd.b8_hi)',
           '[bias_lane] "v"(bias_lane)', '[rs_lane] "v"(rs_lane)', '[voff] "v"(rd.ring.voff)',
           '[src_lo] "s"(src_lo)', '[src_hi] "s"(src_hi)', '[ldsw] "s"(ldsw)', '[wave] "s"(wave)']
    clob = ['"memory"', '"scc"', '"m0"', '"s96"', '"s97"']
    pinned = set(range(OUT, OUT + (8 if full else 2)))
    for ct in range(NCT):
        pinned |= set(range(H + 24 * ct, H + 24 * ct + 4 * nfrag))
    clob += ['"v%d"' % v for v in range(OUT, LAST_VGPR + 1) if v not in pinned]
    clob += ['"a%d"' % a for a in range(N_AGPR)]
    out.append("        : " + ", ".join(outs))
    out.append("        : " + ", ".join(ins))
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    out.append("")


def set_chunk(chunk):
    """ring of 128 KiB in chunks of `chunk` bytes (16 KiB: 8 slots, 32 KiB: 4 slots, half the boundaries)"""
    global CHUNK, SLOTS, LOOK, GPC
    import gen_mx_asm as G
    G.CHUNK = CHUNK = chunk
    G.SLOTS = SLOTS = RING // chunk
    G.LOOK = LOOK = SLOTS - 1
    GPC = CHUNK // (NWAVES * 1024)


def main():
    cfg = {}
    for a in sys.argv[1:]:
        k, v = a.split("=")
        cfg[k] = int(v)
    set_chunk(cfg.get("chunk", CHUNK))
    out = ["// GENERATED by tools/gen_mx2_asm.py %s-- do not edit; see that file for the design." % ("".join(x + " " for x in sys.argv[1:])), ""]
    t = Table(NERF_SHAPES)
    nq = t.first[12]
    units = t.bytes_upto(nq) // 1024
    emit_pass("nerf_full_pass", t, nerf_full_layers(), nq, units, out, cfg)
    nq_s = t.first[9]
    emit_pass("nerf_sigma_pass", t, nerf_full_layers()[:9], nq_s, t.bytes_upto(nq_s) // 1024, out, cfg, full=False)
    print("\n".join(out))


if __name__ == "__main__":
    main()
