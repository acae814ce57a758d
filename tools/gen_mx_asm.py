#!/usr/bin/env python3
"""Generator of the hand-placed instruction streams of the fp16mx (TGTC_PREC_FP16_FP6) hidden layers.

    tools/gen_mx_asm.py nerf  > tgtc-style_amd/csrc/mx_asm_nerf.inc      (the NeRF table of mlp_mx.h, fused ray kernel)
    tools/gen_mx_asm.py bench > tools/microbench/mx_asm_bench.inc          (8 hidden layers, tools/microbench/mx_layer.hip)

What the stream computes is `dense_mx<.., RT = 16, NKB = 2, NPE = 0>` of mlp_mx.h for one or more consecutive 256 -> 256
layers -- the same MFMA sequence per row tile (M0 C1 M1 M2 C2 M3 per 128-deep block, one accumulator chain), the same
epilogue arithmetic (mx_store_act) -- so the results are bit-identical to the HIP loop.  What differs is everything the
compiler decided there: registers are owned by the stream (in-place accumulators, two weight-group buffers, three
accumulator sets), LDS reads run two groups ahead of their use with ONE counted wait per group, the ReLU / split / block
conversion work is placed MFMA gap by MFMA gap, the last row tile's epilogue and the block conversions of a layer run
behind the first MFMAs of the next layer, and no hazard nop or per-MFMA wait is left in the loop.

Ring protocol (mlp_core.h WeightStream PERSIST, mlp_mx.h MxReader::acquire): entering virtual chunk v = wait vmcnt((LOOK-2)
*GPC), s_barrier, issue chunk v+LOOK into the slot of chunk v-1.  A block is entered with the reader's state for DEPTH 1
(group Q0 staged in rd.ub/wb[0], chunks up to chunk_hi(Q0)-1 entered) and left in the same state for group Qend.
"""
import sys

CHUNK = 16384
SLOTS = 8
RING = CHUNK * SLOTS
NWAVES = 8
GPC = CHUNK // (NWAVES * 1024)       # LDS-DMA instructions per wave and chunk
LOOK = SLOTS - 1
KGROUP = 7168


class Table:
    def __init__(self, shapes):
        self.off, self.npe, self.first = [], [], []
        off = 0
        for (rt, nkb, npe) in shapes:
            self.first.append(len(self.off))
            for _ in range(rt):
                ng = nkb + (1 if npe else 0)
                for q in range(ng):
                    n = 0 if q < nkb else npe
                    size = n * 2048 if n else KGROUP
                    if off // RING != (off + size - 1) // RING:
                        off = (off // RING + 1) * RING
                    self.off.append(off)
                    self.npe.append(n)
                    off += size
        self.first.append(len(self.off))
        self.n = len(self.off)
        self.bytes = (off + CHUNK - 1) // CHUNK * CHUNK

    def size(self, q):
        return self.npe[q] * 2048 if self.npe[q] else KGROUP

    def chunk_lo(self, q):
        return self.off[q] // CHUNK

    def chunk_hi(self, q):
        return (self.off[q] + self.size(q) - 1) // CHUNK

    def bytes_upto(self, nq):
        end = self.off[nq - 1] + self.size(nq - 1)
        return (end + CHUNK - 1) // CHUNK * CHUNK


NERF_SHAPES = [(16, 0, 2), (16, 2, 0), (16, 2, 0), (16, 2, 0), (16, 2, 0), (16, 2, 2),
               (16, 2, 0), (16, 2, 0), (1, 2, 0), (16, 2, 0), (8, 2, 1), (1, 1, 0)]
NERF_RT = [16, 16, 16, 16, 16, 16, 16, 16, 1, 16, 8, 1]
BENCH_SHAPES = [(16, 2, 0)] * 8


# ------------------------------------------------------------------------------------------------ register map
# (all even-aligned tuples; v0..v47 stay with the compiler)
class R:
    T = 40          # v40..v43: relu'd accumulator values
    RS = 44         # v44..v46: row-exponent words of the three accumulator sets
    MXK = 47        # packed running maximum of the block being produced
    M0, M1, M2, M3 = 48, 49, 50, 51    # scratch of the block close
    ACC = 52        # three accumulator sets -> v52..v63
    L16 = 64        # staging registers of the lo halves -> v64..v79
    A = 80          # activation set A: h 32 | h6 12 | l6 12 | sc 2 -> v80..v137
    B = 138         # activation set B -> v138..v195
    W0 = 196        # weight group buffer 0: U 16 | Wl6 6 | Wh6 6 -> v196..v223 (the reader's staged group, in/out operand)
    W1 = 224        # weight group buffer 1 -> v224..v251


ACT_H, ACT_H6, ACT_L6, ACT_SC = 0, 32, 44, 56


def vr(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


class Emitter:
    """Collects instructions; tracks LDS operations for counted lgkmcnt waits."""

    def __init__(self):
        self.lines = []
        self.lds_issued = 0          # number of LDS ops issued so far
        self.lds_done = 0            # ops [0, lds_done) are known complete

    def emit(self, s):
        self.lines.append(s)

    def lds(self, s):
        self.emit(s)
        self.lds_issued += 1
        return self.lds_issued - 1   # op id

    def need(self, op):
        """make LDS op `op` (and all older ones) complete"""
        if op is None or op < self.lds_done:
            return
        n = self.lds_issued - 1 - op
        # (lgkmcnt is a 4-bit counter: "at most 15 outstanding" retires everything older than the 15 youngest, LDS returns in order)
        self.emit("s_waitcnt lgkmcnt(%d)" % min(n, 15))
        self.lds_done = op + 1


class BlockGen:
    def __init__(self, table, q0, nlayers, bias0, nq_pass, padc, cfg):
        self.t, self.q0, self.nl, self.bias0 = table, q0, nlayers, bias0
        self.nq, self.padc, self.cfg = nq_pass, padc, cfg
        self.qend = q0 + 32 * nlayers
        self.e = Emitter()
        self.unit_op = {}            # (q, unit) -> LDS op id
        self.unit_chunk = {}         # op id -> chunk it reads from
        # what the DEPTH-1 reader has entered when it arrives at group q0 (MxReader::acquire: lo = max(prev + 1, 1))
        self.entered = max(table.chunk_hi(min(q0, nq_pass - 1)) - 1, 0)
        self.exit_entered = max(table.chunk_hi(min(self.qend, nq_pass - 1)) - 1, 0)

    # ---------------------------------------------------------------- ring
    def boundary(self, v):
        e = self.e
        # reads of the chunk whose slot is re-filled (v-1) must have returned in THIS wave before it joins the barrier
        victims = [op for op, ch in self.unit_chunk.items() if ch <= v - 1 and op >= e.lds_done]
        if victims:
            e.need(max(victims))
        if self.cfg.get("abl_ring"):             # timing experiment: no counted wait, barrier or LDS-DMA
            self.entered = v
            return
        e.emit("s_waitcnt vmcnt(%d)" % ((LOOK - 2) * GPC))
        if not self.cfg.get("abl_barrier"):
            e.emit("s_barrier")
        ch = v + LOOK
        if ch < self.padc:
            lo, hi, off = "%[src_lo]", "%[src_hi]", ch * CHUNK
        else:
            assert ch < self.padc + SLOTS
            lo, hi, off = "%[nxt_lo]", "%[nxt_hi]", (ch - self.padc) * CHUNK
        slot = ch % SLOTS
        e.emit("s_add_u32 s96, %s, 0x%x" % (lo, off))
        e.emit("s_addc_u32 s97, %s, 0" % hi)
        e.emit("s_add_u32 m0, %%[ldsw], 0x%x" % (slot * CHUNK))
        e.emit("s_nop 0")
        for j in range(GPC):
            e.emit("global_load_lds_dwordx4 %%[voff], s[96:97] offset:%d" % (j * 1024))
        self.entered = v

    def readable(self, q):
        return q <= self.qend and q < self.nq

    def acquire_for(self, q):
        """enter the chunks needed to read group q (never beyond what the reader expects to find at the block's exit)"""
        g = min(q, self.qend, self.nq - 1)
        want = self.t.chunk_hi(g) - 1
        assert want <= self.exit_entered
        while self.entered < want:
            self.boundary(self.entered + 1)

    # ---------------------------------------------------------------- weight reads
    def wbase(self, q):
        return R.W0 if ((q - self.q0) & 1) == 0 else R.W1

    def read_unit(self, q, u):
        """issue the LDS read of unit u of group q into its buffer"""
        if not self.readable(q):
            return
        if self.cfg.get("abl_reads") and q > self.q0 + 2 and q < self.qend:
            return                   # timing experiment: no weight reads inside the block
        t, e = self.t, self.e
        assert t.npe[q] == 0, "K groups only"
        assert t.chunk_hi(q) - 1 <= self.entered, "group %d read before its chunks were entered" % q
        o = t.off[q] % RING
        w = self.wbase(q)
        if self.cfg.get("b128") and u >= 4:
            if u == 7:
                return
            piece = {4: 0, 6: 1, 5: 2}[u]
            dst, off, wide = vr(w + 16 + 4 * piece, 4), o + 4096 + 1024 * piece, True
        elif u < 4:
            dst, off, wide = vr(w + 4 * u, 4), o + 1024 * u, True
        elif u == 4:
            dst, off, wide = vr(w + 16, 4), o + 4096, True
        elif u == 5:
            dst, off, wide = vr(w + 22, 4), o + 5120, True
        elif u == 6:
            dst, off, wide = vr(w + 20, 2), o + 6144, False
        else:
            dst, off, wide = vr(w + 26, 2), o + 6144 + 512, False
        if wide:
            base = "%[lane_lo]" if off < 65536 else "%[lane_hi]"
            op = e.lds("ds_read_b128 %s, %s offset:%d" % (dst, base, off % 65536))
        else:
            base = "%[b8_lo]" if off < 65536 else "%[b8_hi]"
            op = e.lds("ds_read_b64 %s, %s offset:%d" % (dst, base, off % 65536))
        self.unit_op[(q, u)] = op
        self.unit_chunk[op] = (t.off[q] + (off - o)) // CHUNK

    # ---------------------------------------------------------------- epilogue pieces (mx_store_act)
    def epi_ops(self, rt, yset, acc):
        """the epilogue of row tile rt as a list of fillers: closures, "ACC_FREE" (the accumulator set has been read)"""
        ops = []
        e = self.e
        ks = rt // 2
        T = R.T
        for i in range(4):
            ops.append(lambda i=i: e.emit("v_max_i32_e32 %s, 0, %s" % (vr(T + i), vr(acc + i))))
        ops.append("ACC_FREE")
        for half in range(2):
            d = (rt & 1) * 2 + half
            h = yset + ACT_H + 4 * ks + d
            l = R.L16 + 4 * (ks & 3) + d
            t0, t1 = T + 2 * half, T + 2 * half + 1
            ops.append(lambda h=h, t0=t0, t1=t1: e.emit("v_cvt_pk_f16_f32 %s, %s, %s" % (vr(h), vr(t0), vr(t1))))
            ops.append(lambda l=l, t0=t0, h=h: e.emit(
                "v_fma_mixlo_f16 %s, %s, 1.0, -%s op_sel:[0,0,0] op_sel_hi:[0,0,1]" % (vr(l), vr(t0), vr(h))))
            if (rt & 7) == 0 and half == 0:
                ops.append(lambda h=h: e.emit("v_mov_b32_e32 %s, %s" % (vr(R.MXK), vr(h))))
            else:
                ops.append(lambda h=h: e.emit("v_pk_max_u16 %s, %s, %s" % (vr(R.MXK), vr(R.MXK), vr(h))))
            ops.append(lambda l=l, t1=t1, h=h: e.emit(
                "v_fma_mixhi_f16 %s, %s, 1.0, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (vr(l), vr(t1), vr(h))))
        if (rt & 7) == 7:
            kb = rt // 8
            m0, m1, m2, m3 = R.M0, R.M1, R.M2, R.M3
            ops.append(lambda: e.emit(
                "v_max_u32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" % (vr(m0), vr(R.MXK), vr(R.MXK))))
            ops.append(lambda: e.emit("v_lshrrev_b32_e32 %s, 10, %s" % (vr(m0), vr(m0))))
            ops.append(lambda: e.emit("v_add_u32_e32 %s, 0x63, %s" % (vr(m1), vr(m0))))      # byte_l = byte_h - 12
            ops.append(lambda: e.emit("v_lshlrev_b32_e32 %s, 23, %s" % (vr(m3), vr(m1))))
            ops.append(lambda: e.emit("v_cvt_scalef32_pk32_fp6_f16 %s, %s, %s" % (
                vr(yset + ACT_L6 + 6 * kb, 6), vr(R.L16, 16), vr(m3))))
            ops.append(lambda: e.emit("v_add_u32_e32 %s, 0x6f, %s" % (vr(m0), vr(m0))))      # byte_h
            ops.append(lambda: e.emit("v_lshlrev_b32_e32 %s, 23, %s" % (vr(m2), vr(m0))))
            ops.append(lambda: e.emit("v_lshl_or_b32 %s, %s, 8, %s" % (vr(yset + ACT_SC + kb), vr(m1), vr(m0))))
            ops.append(lambda: e.emit("v_cvt_scalef32_pk32_fp6_f16 %s, %s, %s" % (
                vr(yset + ACT_H6 + 6 * kb, 6), vr(yset + ACT_H + 16 * kb, 16), vr(m2))))
        return ops

    # ---------------------------------------------------------------- the block
    def generate(self):
        e, cfg = self.e, self.cfg
        dist = cfg.get("dist", 2)
        assert dist in (1, 2)
        total_rt = 16 * self.nl
        accs = [R.ACC, R.ACC + 4, R.ACC + 8]    # row tile r accumulates in set r % 3
        bias_op = {}                            # global row tile -> id of its last bias / row-exponent read

        def load_bias(grt):
            if grt >= total_rt:
                return
            layer, rt = divmod(grt, 16)
            boff = self.bias0 + 256 * layer + 16 * rt
            e.lds("ds_read_b128 %s, %%[bias_lane] offset:%d" % (vr(accs[grt % 3], 4), boff * 4))
            bias_op[grt] = e.lds("ds_read_u16 %s, %%[rs_lane] offset:%d" % (vr(R.RS + grt % 3), boff * 2))

        fillers = []

        def run_filler(grt):
            f = fillers.pop(0)
            if f == "ACC_FREE":
                load_bias(grt + 2)              # the set just read (row tile grt-1) is used next by row tile grt+2
            elif cfg.get("abl_epi"):
                pass                            # timing experiment: no epilogue work
            else:
                n0 = len(e.lines)
                f()
                if cfg.get("abl_cvt") and "v_cvt_scalef32" in e.lines[-1]:
                    del e.lines[n0:]            # timing experiment: no block conversions

        # prologue: biases of the first three row tiles; with dist 2 group q0+1 in a burst (group q0 is staged by the caller)
        for r in range(3):
            load_bias(r)
        if dist == 2:
            self.acquire_for(self.q0 + 1)
            for u in (0, 4, 6, 1, 2, 5, 7, 3):
                self.read_unit(self.q0 + 1, u)   # (b128: unit 7 is part of piece 6)

        for grt in range(total_rt):
            layer, rt = divmod(grt, 16)
            xset = R.A if (layer % 2 == 0) else R.B
            acc, rs = accs[grt % 3], R.RS + grt % 3
            if grt > 0:                         # the previous row tile's epilogue is this row tile's filler stream
                pl, prt = divmod(grt - 1, 16)
                fillers.extend(self.epi_ops(prt, R.B if (pl % 2 == 0) else R.A, accs[(grt - 1) % 3]))
            # a layer's second group is the first reader of what the previous layer's LAST row tile produced
            # (k-step 7, block 1's fp6 operands and scales): everything pending must have been issued by then
            deadline = 6 if (rt == 0 and grt > 0) else 12
            for kb in range(2):
                q = self.q0 + grt * 2 + kb
                w = self.wbase(q)
                if kb == 1 and deadline == 6:
                    while fillers:
                        run_filler(grt)
                self.acquire_for(q + dist)
                ops_needed = [self.unit_op[(q, u)] for u in range(8) if (q, u) in self.unit_op]
                if kb == 0:
                    ops_needed.append(bias_op[grt])
                if ops_needed:
                    e.need(max(ops_needed))     # ONE counted wait per group
                for j, (kind, idx) in enumerate((("M", 0), ("C", 1), ("M", 1), ("M", 2), ("C", 2), ("M", 3))):
                    gap = kb * 6 + j
                    if kind == "M":
                        e.emit("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (
                            vr(acc, 4), vr(w + 4 * idx, 4), vr(xset + ACT_H + 4 * (4 * kb + idx), 4), vr(acc, 4)))
                        units = [idx]
                    elif idx == 1:
                        e.emit("v_mfma_scale_f32_16x16x128_f8f6f4 %s, %s, %s, %s, %s, %s op_sel:[1,0,0] op_sel_hi:[0,0,0] cbsz:2 blgp:2" % (
                            vr(acc, 4), vr(w + 16, 6), vr(xset + ACT_H6 + 6 * kb, 6), vr(acc, 4), vr(rs), vr(xset + ACT_SC + kb)))
                        units = [4] if cfg.get("b128") else [4, 6]
                    else:
                        e.emit("v_mfma_scale_f32_16x16x128_f8f6f4 %s, %s, %s, %s, %s, %s op_sel:[0,1,0] op_sel_hi:[0,0,0] cbsz:2 blgp:2" % (
                            vr(acc, 4), vr(w + 22, 6), vr(xset + ACT_L6 + 6 * kb, 6), vr(acc, 4), vr(rs), vr(xset + ACT_SC + kb)))
                        units = [6, 5] if cfg.get("b128") else [5, 7]
                    for u in units:             # refill: the same units of the group `dist` ahead
                        self.read_unit(q + dist, u)
                    # the previous row tile's accumulator may be read from gap 2 on (three MFMAs behind its last one)
                    if gap >= 2 and fillers:
                        remaining = max(deadline - 1 - gap, 0) + 1
                        n = (len(fillers) + remaining - 1) // remaining
                        if deadline == 12:
                            n = min(n, cfg.get("max_fill", 3))
                        for _ in range(n):
                            if fillers:
                                run_filler(grt)
        # tail: the last row tile's epilogue has nothing to hide behind inside this block
        e.emit("s_nop 7")
        e.emit("s_nop 7")
        pl, prt = divmod(total_rt - 1, 16)
        fillers.extend(self.epi_ops(prt, R.B if (pl % 2 == 0) else R.A, accs[(total_rt - 1) % 3]))
        while fillers:
            run_filler(total_rt)
        assert self.entered == self.exit_entered, (self.entered, self.exit_entered)
        e.emit("s_waitcnt lgkmcnt(0)")          # the reader's staged group (buffer 0) is complete when the compiler takes over
        return e.lines


# ================================================================================================ the whole pass
# All twelve layers of the fp16mx NeRF pass (mlp_nerf_mx_chain.h nerf_chain_mx, FULL) as ONE stream: encoding groups (fp16
# hi / lo k-steps, three products each), the skip layer, the sigma head, base_remap, the colour head.  Nothing of the
# compiler's is live across it but the per-lane addresses; the encodings come in pinned (v16..v39), sigma and the colour
# head's accumulator go out in v12..v15.
class Layer:
    def __init__(self, name, rt, nkb, npe, src, sink, pe):
        self.name, self.rt, self.nkb, self.npe, self.src, self.sink, self.pe = name, rt, nkb, npe, src, sink, pe


KEEP = {"pe": (16, 24), "dir": (32, 36)}     # (hi base, lo base): Ph[k] = v[16+4k..], Pl[k] = v[24+4k..]; Dh = v[32..35], Dl = v[36..39]
OUT_SIG, OUT_RGB = 12, 13                     # v12: sigma; v13..v15: the colour head's accumulator rows 0..2


def nerf_full_layers():
    A, B = R.A, R.B
    return [Layer("L0", 16, 0, 2, None, A, "pe"), Layer("L1", 16, 2, 0, A, B, None), Layer("L2", 16, 2, 0, B, A, None),
            Layer("L3", 16, 2, 0, A, B, None), Layer("L4", 16, 2, 0, B, A, None), Layer("L5", 16, 2, 2, A, B, "pe"),
            Layer("L6", 16, 2, 0, B, A, None), Layer("L7", 16, 2, 0, A, B, None), Layer("SIG", 1, 2, 0, B, "sigma", None),
            Layer("REMAP", 16, 2, 0, B, A, None), Layer("C0", 8, 2, 1, A, B, "dir"), Layer("C1", 1, 1, 0, B, "rgb", None)]


class PassGen(BlockGen):
    def __init__(self, table, layers, nq_pass, padc, cfg):
        BlockGen.__init__(self, table, 0, 0, 0, nq_pass, padc, cfg)
        self.layers = layers
        self.qend = nq_pass
        self.exit_entered = max(table.chunk_hi(nq_pass - 1) - 1, 0)
        self.entered = 0                                     # enter<0, NQ>() has run the boundary of virtual chunk 0

    def readable(self, q):
        return q < self.nq

    def acquire_for(self, q):
        g = min(q, self.nq - 1)
        want = self.t.chunk_hi(g) - 1
        while self.entered < want:
            self.boundary(self.entered + 1)

    # units of a group: (unit id, register offset in the buffer, registers, byte offset in the group, wide)
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
        o = t.off[q] % RING
        w = self.wbase(q)
        off = o + boff
        if wide:
            base = "%[lane_lo]" if off < 65536 else "%[lane_hi]"
            op = e.lds("ds_read_b128 %s, %s offset:%d" % (vr(w + roff, 4), base, off % 65536))
        else:
            base = "%[b8_lo]" if off < 65536 else "%[b8_hi]"
            op = e.lds("ds_read_b64 %s, %s offset:%d" % (vr(w + roff, 2), base, off % 65536))
        self.unit_op[(q, u)] = op
        self.unit_chunk[op] = (t.off[q] + boff) // CHUNK

    def mfmas(self, q, layer, acc, rs):
        """the MFMA instructions of group q with, for each, the buffer registers it is the LAST reader of"""
        w = self.wbase(q)
        n = self.t.npe[q]
        out = []
        f16 = "v_mfma_f32_16x16x32_f16 %s, %s, %s, %s"
        if n:
            hi0, lo0 = KEEP[layer.pe]
            for k in range(n):
                ph, pl = vr(hi0 + 4 * k, 4), vr(lo0 + 4 * k, 4)
                uh, ul = w + 8 * k, w + 8 * k + 4
                out.append((f16 % (vr(acc, 4), vr(uh, 4), ph, vr(acc, 4)), []))
                out.append((f16 % (vr(acc, 4), vr(ul, 4), ph, vr(acc, 4)), list(range(8 * k + 4, 8 * k + 8))))
                out.append((f16 % (vr(acc, 4), vr(uh, 4), pl, vr(acc, 4)), list(range(8 * k, 8 * k + 4))))
            return out
        # which 128-deep block of the source: the group's position among the row tile's K groups
        kb = self.group_kb[q]
        x = layer.src
        sc = "op_sel:[%d,%d,0] op_sel_hi:[0,0,0] cbsz:2 blgp:2"
        fp6 = "v_mfma_scale_f32_16x16x128_f8f6f4 %s, %s, %s, %s, %s, %s " + sc
        for j, (kind, idx) in enumerate((("M", 0), ("C", 1), ("M", 1), ("M", 2), ("C", 2), ("M", 3))):
            if kind == "M":
                out.append((f16 % (vr(acc, 4), vr(w + 4 * idx, 4), vr(x + ACT_H + 4 * (4 * kb + idx), 4), vr(acc, 4)),
                            list(range(4 * idx, 4 * idx + 4))))
            elif idx == 1:
                out.append((fp6 % (vr(acc, 4), vr(w + 16, 6), vr(x + ACT_H6 + 6 * kb, 6), vr(acc, 4), vr(rs), vr(x + ACT_SC + kb), 1, 0),
                            list(range(16, 22))))
            else:
                out.append((fp6 % (vr(acc, 4), vr(w + 22, 6), vr(x + ACT_L6 + 6 * kb, 6), vr(acc, 4), vr(rs), vr(x + ACT_SC + kb), 0, 1),
                            list(range(22, 28))))
        return out

    def generate(self):
        e, cfg, t = self.e, self.cfg, self.t
        dist = 2
        accs = [R.ACC, R.ACC + 4, R.ACC + 8]
        # the groups in stream order, with their layer / row tile / role
        groups, self.group_kb = [], {}
        rows = []                                  # global row tiles: (layer index, rt, [groups])
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
            e.lds("ds_read_b128 %s, %%[bias_lane] offset:%d" % (vr(accs[grt % 3], 4), boff * 4))
            if L.nkb:
                bias_op[grt] = e.lds("ds_read_u16 %s, %%[rs_lane] offset:%d" % (vr(R.RS + grt % 3), boff * 2))
            else:
                bias_op[grt] = e.lds_issued - 1

        fillers = []
        tail_of = [None]                           # (set, block) the pending fillers still have to complete

        def run_filler(grt):
            f = fillers.pop(0)
            if isinstance(f, tuple) and f[0] == "ACC_FREE":
                load_bias(f[1] + 3)                  # the set row f[1] accumulated in is used next by row f[1] + 3
            else:
                f()

        def flush(grt, gap=99):
            # the fillers start by reading the previous row tile's accumulator: fewer than three MFMAs behind its last one
            # (a one-block layer read by the very next row tile: C0 -> C1) the result is not there yet -- drain explicitly
            if fillers and gap < 3:
                e.emit("s_nop 7")
                e.emit("s_nop 7")
            while fillers:
                run_filler(grt)
            tail_of[0] = None

        def epilogue(grt):
            """fillers of row tile grt's epilogue"""
            li, rt, _ = rows[grt]
            L = self.layers[li]
            acc = accs[grt % 3]
            if L.sink == "sigma":
                return [lambda: e.emit("v_mov_b32_e32 %s, %s" % (vr(OUT_SIG), vr(acc))), ("ACC_FREE", grt)]
            if L.sink == "rgb":
                return [lambda i=i: e.emit("v_mov_b32_e32 %s, %s" % (vr(OUT_RGB + i), vr(acc + i))) for i in range(3)] + [("ACC_FREE", grt)]
            return [("ACC_FREE", grt) if f == "ACC_FREE" else f for f in self.epi_ops(rt, L.sink, acc)]

        for r in range(3):
            load_bias(r)
        self.acquire_for(1)
        for unit in self.units(1):
            self.read_unit_at(1, unit)

        free = {0: set(), 1: set()}                # buffer parity -> buffer registers no MFMA of the running group needs any more
        pending = {}                               # group -> units still to read
        for grt, (li, rt, gs) in enumerate(rows):
            L = self.layers[li]
            acc, rs = accs[grt % 3], R.RS + grt % 3
            if grt > 0:
                pli, prt, _ = rows[grt - 1]
                PL = self.layers[pli]
                fillers.extend(epilogue(grt - 1))
                if isinstance(PL.sink, int):
                    tail_of[0] = (PL.sink, prt // 8)
            n_row = sum(3 * t.npe[g] if t.npe[g] else 6 for g in gs)
            gap = 0
            for g in gs:
                # a K group that reads the block the pending fillers are still producing: they run first
                if t.npe[g] == 0 and tail_of[0] == (L.src, self.group_kb[g]):
                    flush(grt, gap)
                if t.npe[g] == 0 and fillers and tail_of[0] is not None and tail_of[0][0] == L.src and L.nkb == 1:
                    flush(grt, gap)
                self.acquire_for(g + dist)
                ops_needed = [self.unit_op[(g, u[0])] for u in self.units(g) if (g, u[0]) in self.unit_op]
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
                pending = list(self.units(nxt)) if self.readable(nxt) else []

                def issue_ready():
                    for unit in list(pending):
                        if set(range(unit[1], unit[1] + unit[2])) <= free[par]:
                            self.read_unit_at(nxt, unit)
                            pending.remove(unit)
                issue_ready()
                for text, released in self.mfmas(g, L, acc, rs):
                    e.emit(text)
                    free[par] |= set(released)
                    issue_ready()
                    if gap >= 2 and fillers:
                        # deadline: the first later group of this row tile that will flush (same rule as above), else the row's end
                        remaining = max(n_row - 1 - gap, 0) + 1
                        if tail_of[0] is not None and tail_of[0][0] == L.src:
                            first_dep = 0
                            seen = 0
                            for gg in gs:
                                cnt = 3 * t.npe[gg] if t.npe[gg] else 6
                                if t.npe[gg] == 0 and self.group_kb[gg] == tail_of[0][1]:
                                    first_dep = seen
                                    break
                                seen += cnt
                            remaining = max(first_dep - 1 - gap, 0) + 1
                        n = (len(fillers) + remaining - 1) // remaining     # a row's epilogue always finishes within the next row
                        for _ in range(n):
                            if fillers:
                                run_filler(grt)
                    gap += 1
                assert not pending, "units of group %d could not be placed" % nxt
            if not fillers:
                tail_of[0] = None
        # the colour head's accumulator: drained, then exported
        e.emit("s_nop 7")
        e.emit("s_nop 7")
        fillers.extend(epilogue(total_rt - 1))
        flush(total_rt)
        assert self.entered == self.exit_entered, (self.entered, self.exit_entered)
        e.emit("s_waitcnt lgkmcnt(0)")
        return e.lines


def emit_pass(name, table, layers, nq_pass, units_pass, cfg, out):
    nchunk = (units_pass + 15) // 16
    padc = (nchunk + SLOTS - 1) // SLOTS * SLOTS
    gen = PassGen(table, layers, nq_pass, padc, cfg)
    lines = gen.generate()
    stats = {}
    for l in lines:
        k = l.split()[0]
        stats[k] = stats.get(k, 0) + 1
    out.append("// pass %s: %d groups, %d layers; %d instructions; %s" % (
        name, nq_pass, len(layers), len(lines), ", ".join("%s %d" % kv for kv in sorted(stats.items(), key=lambda kv: -kv[1])[:12])))
    out.append("// keep[0..5] = Ph[0], Ph[1], Pl[0], Pl[1], Dh, Dl (the encodings' fp16 hi / lo B fragments) in v16..v39; sigma (lanes 0..15)")
    out.append("// and rows 0..2 of the colour head's accumulator come back in v12..v15.  The reader must have run enter<0, NQ>().")
    out.append("template <class Reader>")
    out.append("__device__ __forceinline__ void mx_asm_%s(Reader& rd, lds_cptr bias_lane, lds_cptr rs_lane, half8 (&keep)[6], float& sigma, float (&rgb)[3]) {" % name)
    out.append("    static_assert(Reader::DEPTH == 1 && Reader::Ring::STAG == 0, \"the asm streams take over a DEPTH-1 reader\");")
    out.append("    const unsigned src_lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)rd.ring.src[0]), src_hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)rd.ring.src[0] >> 32));")
    out.append("    const unsigned nxt_lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)rd.ring.next), nxt_hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)rd.ring.next >> 32));")
    out.append("    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(size_t)TGTC_LPTR(rd.ring.lds_wave));")
    out.append("    v16u wu = mx_cat4(rd.ub[0][0], rd.ub[0][1], rd.ub[0][2], rd.ub[0][3]);")
    out.append("    u6v wl = rd.wb[0][0], wh = rd.wb[0][1];")

    def rg(base, n):
        return "{v[%d:%d]}" % (base, base + n - 1)
    outs = ['"=&{v%d}"(sigma)' % OUT_SIG] + ['"=&{v%d}"(rgb[%d])' % (OUT_RGB + i, i) for i in range(3)]
    outs += ['"+%s"(wu)' % rg(R.W0, 16), '"+%s"(wl)' % rg(R.W0 + 16, 6), '"+%s"(wh)' % rg(R.W0 + 22, 6)]
    outs += ['"+%s"(keep[%d])' % (rg(16 + 4 * i, 4), i) for i in range(6)]
    out.append("    asm volatile(")
    out.append(block_text(lines))
    ins = ['[lane_lo] "v"(rd.ring.lane_lo)', '[lane_hi] "v"(rd.ring.lane_hi)', '[b8_lo] "v"(rd.b8_lo)', '[b8_hi] "v"(rd.b8_hi)',
           '[bias_lane] "v"(bias_lane)', '[rs_lane] "v"(rs_lane)', '[voff] "v"(rd.ring.voff)',
           '[src_lo] "s"(src_lo)', '[src_hi] "s"(src_hi)', '[nxt_lo] "s"(nxt_lo)', '[nxt_hi] "s"(nxt_hi)', '[ldsw] "s"(ldsw)']
    clob = ['"memory"', '"scc"', '"m0"', '"s96"', '"s97"']
    owned = list(range(R.T, R.W0)) + list(range(R.W1, R.W1 + 28))
    clob += ['"v%d"' % v for v in owned]
    out.append("        : " + ", ".join(outs))
    out.append("        : " + ", ".join(ins))
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    out.append("")


def block_text(lines):
    return "\n".join('        "%s\\n\\t"' % l for l in lines)


def emit_block(name, table, q0, nlayers, bias0, nq_pass, units_pass, cfg, out):
    nchunk = (units_pass + 15) // 16
    padc = (nchunk + SLOTS - 1) // SLOTS * SLOTS
    gen = BlockGen(table, q0, nlayers, bias0, nq_pass, padc, cfg)
    lines = gen.generate()
    final_b = (nlayers % 2 == 1)
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    stats = {}
    for l in lines:
        k = l.split()[0]
        stats[k] = stats.get(k, 0) + 1
    out.append("// block %s: groups [%d, %d), %d layers, bias0 %d; %d instructions, %d MFMA; %s" % (
        name, q0, q0 + 32 * nlayers, nlayers, bias0, len(lines), n_mfma,
        ", ".join("%s %d" % kv for kv in sorted(stats.items(), key=lambda kv: -kv[1])[:12])))
    out.append("template <class Reader>")
    out.append("// keep[0..5]: 24 registers of the caller (the encodings' hi / lo k-steps in the NeRF chain) pinned to v16..v39")
    out.append("// across the block: as plain live values hipcc spills them around the block's clobber list, and a scratch reload waits vmcnt(0),")
    out.append("// i.e. for the whole look-ahead of the ring")
    out.append("__device__ __forceinline__ void mx_asm_%s(Reader& rd, lds_cptr bias_lane, lds_cptr rs_lane, const MxAct<2>& X, MxAct<2>& Y, half8 (&keep)[6]) {" % name)
    out.append("    static_assert(Reader::DEPTH == 1 && Reader::Ring::STAG == 0, \"the asm streams hand over a DEPTH-1 reader\");")
    out.append("    MxAsmRegs r;")
    out.append("    mx_asm_load(r, X, rd);")
    out.append("    const unsigned src_lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)rd.ring.src[0]), src_hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)rd.ring.src[0] >> 32));")
    out.append("    const unsigned nxt_lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)rd.ring.next), nxt_hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)rd.ring.next >> 32));")
    out.append("    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(size_t)TGTC_LPTR(rd.ring.lds_wave));")
    a, b = R.A, R.B

    def rg(base, n):
        return "{v[%d:%d]}" % (base, base + n - 1)
    xs = [("r.h0", a + ACT_H, 16), ("r.h1", a + ACT_H + 16, 16), ("r.h6a", a + ACT_H6, 6), ("r.h6b", a + ACT_H6 + 6, 6),
          ("r.l6a", a + ACT_L6, 6), ("r.l6b", a + ACT_L6 + 6, 6), ("r.sc", a + ACT_SC, 2)]
    outs = []
    if final_b:
        out.append("    MxAsmRegs y;")
        outs += ['"=%s"(%s)' % (rg(base - a + b, n), nm.replace("r.", "y.")) for nm, base, n in xs]
    outs += ['"+%s"(%s)' % (rg(base, n), nm) for nm, base, n in xs]
    outs += ['"+%s"(r.wu)' % rg(R.W0, 16), '"+%s"(r.wl)' % rg(R.W0 + 16, 6), '"+%s"(r.wh)' % rg(R.W0 + 22, 6)]
    outs += ['"+%s"(keep[%d])' % (rg(16 + 4 * i, 4), i) for i in range(6)]
    out.append("    asm volatile(")
    out.append(block_text(lines))
    ins = ['[lane_lo] "v"(rd.ring.lane_lo)', '[lane_hi] "v"(rd.ring.lane_hi)', '[b8_lo] "v"(rd.b8_lo)', '[b8_hi] "v"(rd.b8_hi)',
           '[bias_lane] "v"(bias_lane)', '[rs_lane] "v"(rs_lane)', '[voff] "v"(rd.ring.voff)',
           '[src_lo] "s"(src_lo)', '[src_hi] "s"(src_hi)', '[nxt_lo] "s"(nxt_lo)', '[nxt_hi] "s"(nxt_hi)', '[ldsw] "s"(ldsw)']
    clob = ['"memory"', '"scc"', '"m0"', '"s96"', '"s97"']
    owned = list(range(R.T, R.A))                      # temps, accumulators, staging
    if not final_b:
        owned += list(range(R.B, R.B + 58))
    owned += list(range(R.W1, R.W1 + 28))
    clob += ['"v%d"' % v for v in owned]
    out.append("        : " + ", ".join(outs))
    out.append("        : " + ", ".join(ins))
    out.append("        : " + ", ".join(clob) + ");")
    if final_b:
        out.append("    r.h0 = y.h0, r.h1 = y.h1, r.h6a = y.h6a, r.h6b = y.h6b, r.l6a = y.l6a, r.l6b = y.l6b, r.sc = y.sc;")
    out.append("    mx_asm_store(r, Y, rd);")
    out.append("}")
    out.append("")


def main():
    global CHUNK, SLOTS, GPC, LOOK
    which = sys.argv[1] if len(sys.argv) > 1 else "bench"
    cfg = {"dist": 2, "max_fill": 3, "b128": 1}      # b128: the stream layout since round 4 (seven reads per K group)
    for a in sys.argv[2:]:
        k, v = a.split("=")
        cfg[k] = int(v)
    if "chunk" in cfg:
        CHUNK = cfg["chunk"]
        SLOTS = RING // CHUNK
        GPC = CHUNK // (NWAVES * 1024)
        LOOK = SLOTS - 1
    out = ["// GENERATED by tools/gen_mx_asm.py %s -- do not edit; see that file for the design." % " ".join(sys.argv[1:]), ""]
    if which == "bench":
        t = Table(BENCH_SHAPES)
        units = t.bytes_upto(t.n) // 1024
        emit_block("bench_a", t, 0, 4, 0, t.n, units, cfg, out)
        emit_block("bench_b", t, 128, 4, 1024, t.n, units, cfg, out)
    else:
        t = Table(NERF_SHAPES)
        bias0 = [0]
        for rt in NERF_RT:
            bias0.append(bias0[-1] + 16 * rt)
        for full in (True, False):
            nq = t.first[12] if full else t.first[9]
            units = t.bytes_upto(nq) // 1024
            tag = "full" if full else "sigma"
            emit_block("nerf_%s_l1_4" % tag, t, t.first[1], 4, bias0[1], nq, units, cfg, out)
            emit_block("nerf_%s_l6_7" % tag, t, t.first[6], 2, bias0[6], nq, units, cfg, out)
            if full:
                emit_block("nerf_%s_l9" % tag, t, t.first[9], 1, bias0[9], nq, units, cfg, out)
                emit_pass("nerf_full_pass", t, nerf_full_layers(), nq, units, cfg, out)
    print("\n".join(out))


if __name__ == "__main__":
    main()
