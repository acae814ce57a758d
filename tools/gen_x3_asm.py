#!/usr/bin/env python3
"""Generator of the fp16x3 (TGTC_PREC_FP16X3) NeRF trunk + sigma head on TWO column tiles per wave, one wave per SIMD.

    tools/gen_x3_asm.py > tgtc-style_amd/csrc/x3_asm_nerf.inc            (per-sample kernel csrc/mlp_nerf_x3s.hip)

The coarse pass of a render (128 depths per ray, sigma only) is the other half of the frame (profiles/r4_kernel_variants.md
section 4).  Its one-tile HIP loop (mlp_core.h dense_layer, eight waves per CU) reaches 0.58 of the matrix pipe; like the
fp16mx loop it reads every weight fragment once per 16 samples and wave.  This is the same move as tools/gen_mx2_asm.py for the
three-product arithmetic: two column tiles per wave (half the LDS bytes and half the weight-side instructions per MFMA), which
needs one wave per SIMD and the accumulator half of the register file, so the whole pass is one generated stream:

  per 32-deep k-step and column tile:  acc += Wh.Ah;  acc += Wl.Ah;  acc += Wh.Al     (dense_layer's order: the same bits)

  VGPR  v16..v17   sigma[ct] (outputs)                      AGPR  a0..a63     eight weight-fragment slots (hi 4 | lo 4 registers;
        v24..v31   ReLU'd accumulator values, 4 per tile                      ds_read straight into them; fragment f lives in slot
        v32..v35   lo halves of the pair being split, 2 per tile              f mod 8 and is read eight fragments ahead of its use)
        v36..v59   three accumulator sets x two tiles              a64..a191   lo halves of the activations [set][tile][32] (B operand
        v60..v187  hi halves of the activations [set][tile][32]               of the third product; written by v_accvgpr_write)
                   (at entry: the encodings, copied to a192.. first)  a192..a223  the point encoding's hi / lo B fragments per tile

Ring protocol: as tools/gen_mx2_asm.py (four waves, four LDS-DMA instructions per wave and 16 KiB chunk, one per MFMA gap; entry,
boundaries and the walk to the padded end inside the stream; the look-ahead runs into the same stream again).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_mx_asm import CHUNK, LOOK, SLOTS, Emitter  # noqa: E402

NWAVES = 4
GPC = CHUNK // (NWAVES * 1024)
NCT = 2
FRAG = 2048
FPC = CHUNK // FRAG
NSLOT = 8                      # weight-fragment register slots = the read-ahead distance in fragments

OUT = 16
T = 24
LT = 32
ACC = 36
H = 60
ACC2 = 188                     # the correction products' accumulators (two_acc), three sets x two tiles -> v188..v211
AW = 0
ALO = 64
AKEEP = 192
N_AGPR = 224
LAST_VGPR = 211


def V(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def A(base, n=1):
    return "a%d" % base if n == 1 else "a[%d:%d]" % (base, base + n - 1)


def acc(s, ct):
    return ACC + (s * NCT + ct) * 4


def acc2(s, ct):
    return ACC2 + (s * NCT + ct) * 4


def hreg(st, ct):
    return H + (st * NCT + ct) * 32


def lreg(st, ct):
    return ALO + (st * NCT + ct) * 32


class Layer:
    """ks: k-steps of the layer; the first `act` of them read activation set `src`, the rest the point encoding"""

    def __init__(self, name, rt, act, npe, src, sink):
        self.name, self.rt, self.act, self.npe, self.src, self.sink = name, rt, act, npe, src, sink
        self.ks = act + npe


def nerf_sigma_layers():
    X, Y = 0, 1
    return [Layer("L0", 16, 0, 2, None, Y), Layer("L1", 16, 8, 0, Y, X), Layer("L2", 16, 8, 0, X, Y), Layer("L3", 16, 8, 0, Y, X),
            Layer("L4", 16, 8, 0, X, Y), Layer("L5", 16, 8, 2, Y, X), Layer("L6", 16, 8, 0, X, Y), Layer("L7", 16, 8, 0, Y, X),
            Layer("SIG", 1, 8, 0, X, "sigma")]


class PassX3:
    def __init__(self, layers, cfg=None):
        self.layers = layers
        self.cfg = cfg or {}
        # two_acc=1 (experiment, measured 4 % SLOWER, profiles/r4_kernel_variants.md): the hi.hi products and the two correction
        # products of a row tile accumulate in separate registers, summed in the epilogue, and the six MFMAs of a k-step are ordered
        # lh_a lh_b hh_a hl_a hl_b hh_b, so that every accumulator is written at most every third MFMA.  Default: ONE chain per
        # tile in dense_layer's order -- bit-identical to the one-tile kernels.
        self.two_acc = self.cfg.get("two_acc", 0)
        self.nfrag = sum(L.rt * L.ks for L in layers)
        self.nchunk = (self.nfrag + FPC - 1) // FPC
        self.padc = (self.nchunk + SLOTS - 1) // SLOTS * SLOTS
        self.e = Emitter()
        self.frag_op = {}            # fragment -> id of its last LDS read
        self.op_chunk = {}           # LDS op -> chunk it reads
        self.dma_q = []
        self.dma_n = self.dma_tick = 0
        self.entered = -1

    # ------------------------------------------------------------------------------------------ ring
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

    def boundary(self, v):
        e = self.e
        self.drain_dma(99)
        victims = [op for op, ch in self.op_chunk.items() if ch <= v - 1 and op >= e.lds_done]
        if victims:
            e.need(max(victims))
        if self.cfg.get("abl_ring"):
            self.entered = v
            return
        e.emit("s_waitcnt vmcnt(%d)" % ((LOOK - 2) * GPC))
        e.emit("s_barrier")
        ch = (v + LOOK) % self.padc
        slot = (v + LOOK) % SLOTS
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
            # (as tools/gen_mx2_asm.py: wave w issues its four pieces at its own place after the boundary, the others branch over it)
            self.dma_n += 1
            for w in range(NWAVES):
                lab = ".Ldmx_%d_%d_%%=" % (self.dma_n, w)
                self.dma_q.append(["s_cmp_eq_u32 %%[wave], %d" % w, "s_cbranch_scc0 %s" % lab] + loads + [lab + ":"])
        else:
            unit = []
            for ln in loads:
                unit.append(ln)
                if ln.startswith("global_load"):
                    self.dma_q.append(unit)
                    unit = []
        self.entered = v

    def acquire_for(self, f):
        """enter the chunk fragment f lies in (chunks up to entered + 1 are readable)"""
        want = min(f, self.nfrag - 1) // FPC - 1
        while self.entered < want:
            self.boundary(self.entered + 1)

    def read_frag(self, f):
        if f >= self.nfrag:
            return
        e = self.e
        self.acquire_for(f)
        if self.cfg.get("abl_reads") and f >= NSLOT:
            self.frag_op[f] = None
            return
        ch = f // FPC
        assert ch - 1 <= self.entered
        off = (ch % SLOTS) * CHUNK + (f % FPC) * FRAG
        w = AW + 8 * (f % NSLOT)
        for part in range(2):
            o = off + part * 1024
            base = "%[lane_lo]" if o < 65536 else "%[lane_hi]"
            op = e.lds("ds_read_b128 %s, %s offset:%d" % (A(w + 4 * part, 4), base, o % 65536))
            self.op_chunk[op] = ch
        self.frag_op[f] = op

    # ------------------------------------------------------------------------------------------ epilogue (store_act, SPLIT)
    def epi_ops(self, rt, st, s):
        e = self.e
        ks = rt // 2
        head, body = [], [[], []]
        for ct in range(NCT):
            a, t = acc(s, ct), T + 4 * ct
            if self.two_acc:
                c = acc2(s, ct)
                for i in (0, 2):
                    head.append(lambda i=i, a=a, t=t, c=c: e.emit("v_pk_add_f32 %s, %s, %s" % (V(t + i, 2), V(a + i, 2), V(c + i, 2))))
                for i in range(4):
                    head.append(lambda i=i, t=t: e.emit("v_max_i32_e32 %s, 0, %s" % (V(t + i), V(t + i))))
            else:
                for i in range(4):
                    head.append(lambda i=i, a=a, t=t: e.emit("v_max_i32_e32 %s, 0, %s" % (V(t + i), V(a + i))))
            for half in range(2):
                d = (rt & 1) * 2 + half
                h = hreg(st, ct) + 4 * ks + d
                lo = LT + 2 * ct + half
                alo = lreg(st, ct) + 4 * ks + d
                t0, t1 = t + 2 * half, t + 2 * half + 1
                ops = body[ct]
                ops.append(lambda h=h, t0=t0, t1=t1: e.emit("v_cvt_pk_f16_f32 %s, %s, %s" % (V(h), V(t0), V(t1))))
                ops.append(lambda lo=lo, t0=t0, h=h: e.emit(
                    "v_fma_mixlo_f16 %s, %s, 1.0, -%s op_sel:[0,0,0] op_sel_hi:[0,0,1]" % (V(lo), V(t0), V(h))))
                ops.append(lambda lo=lo, t1=t1, h=h: e.emit(
                    "v_fma_mixhi_f16 %s, %s, 1.0, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (V(lo), V(t1), V(h))))
                ops.append(lambda alo=alo, lo=lo: e.emit("v_accvgpr_write_b32 %s, %s" % (A(alo), V(lo))))
        merged = []
        for i in range(max(len(body[0]), len(body[1]))):
            for ct in range(NCT):
                if i < len(body[ct]):
                    merged.append(body[ct][i])
        return head + ["ACC_FREE"] + merged

    # ------------------------------------------------------------------------------------------ the pass
    def generate(self):
        e = self.e
        rows = []                    # (layer index, rt, first fragment)
        f = 0
        bias0 = 0
        for li, L in enumerate(self.layers):
            L.bias0 = bias0
            bias0 += 16 * L.rt
            for rt in range(L.rt):
                rows.append((li, rt, f))
                f += L.ks
        assert f == self.nfrag
        total_rt = len(rows)
        bias_op = {}

        def load_bias(grt):
            if grt >= total_rt:
                return
            li, rt, _ = rows[grt]
            boff = self.layers[li].bias0 + 16 * rt
            for ct in range(NCT):
                bias_op[grt] = e.lds("ds_read_b128 %s, %%[bias_lane] offset:%d" % (V(acc(grt % 3, ct), 4), boff * 4))

        fillers = []
        tail_of = [None]             # (set, k-step) the pending fillers still have to complete

        def run_filler():
            fl = fillers.pop(0)
            if isinstance(fl, tuple) and fl[0] == "ACC_FREE":
                load_bias(fl[1] + 3)
            elif self.cfg.get("abl_epi"):
                pass
            else:
                fl()

        def flush(gap=99):
            if fillers and gap < 4:
                e.emit("s_nop 7")
                e.emit("s_nop 7")
            while fillers:
                run_filler()
            tail_of[0] = None

        def epilogue(grt):
            li, rt, _ = rows[grt]
            L = self.layers[li]
            s = grt % 3
            if L.sink == "sigma" and self.two_acc:
                return [lambda ct=ct: e.emit("v_add_f32_e32 %s, %s, %s" % (V(OUT + ct), V(acc(s, ct)), V(acc2(s, ct)))) for ct in range(NCT)] + [("ACC_FREE", grt)]
            if L.sink == "sigma":
                return [lambda ct=ct: e.emit("v_mov_b32_e32 %s, %s" % (V(OUT + ct), V(acc(s, ct)))) for ct in range(NCT)] + [("ACC_FREE", grt)]
            return [("ACC_FREE", grt) if fl == "ACC_FREE" else fl for fl in self.epi_ops(rt, L.sink, s)]

        # ---- prologue: the encodings to their AGPRs, enter the pass, biases of three row tiles, the first NSLOT fragments
        for ct in range(NCT):
            for i in range(16):
                e.emit("v_accvgpr_write_b32 %s, %s" % (A(AKEEP + 16 * ct + i), V(H + 16 * ct + i)))
        self.boundary(0)
        self.drain_dma(99)
        for r in range(3):
            load_bias(r)
        for f0 in range(NSLOT):
            self.read_frag(f0)
            self.drain_dma(99)

        f16 = "v_mfma_f32_16x16x32_f16 %s, %s, %s, %s"
        for grt, (li, rt, f0) in enumerate(rows):
            L = self.layers[li]
            s = grt % 3
            if grt > 0:
                pli, prt, _ = rows[grt - 1]
                PL = self.layers[pli]
                fillers.extend(epilogue(grt - 1))
                if isinstance(PL.sink, int):
                    tail_of[0] = (PL.sink, prt // 2)
            n_row = L.ks * 3 * NCT
            gap = 0
            for ks in range(L.ks):
                fr = f0 + ks
                is_act = ks < L.act
                if is_act and tail_of[0] == (L.src, ks):
                    flush(gap)
                # one counted wait per PAIR of fragments: with eight fragments read ahead the next one landed long ago, and a wait
                # is an instruction the lone wave has to issue like any other (wait_pairs=0: one per fragment, 0.19 per MFMA)
                ops_needed = [self.frag_op[fr]] if self.frag_op.get(fr) is not None else []
                if self.cfg.get("wait_pairs", 1) and fr % 2 == 0 and self.frag_op.get(fr + 1) is not None:
                    ops_needed.append(self.frag_op[fr + 1])
                if ks == 0:
                    ops_needed.append(bias_op[grt])
                if ops_needed:
                    e.need(max(ops_needed))
                w = AW + 8 * (fr % NSLOT)
                # (product, tile) in issue order; products: 0 = Wh.Ah, 1 = Wl.Ah, 2 = Wh.Al
                order = [(1, 0), (1, 1), (0, 0), (2, 0), (2, 1), (0, 1)] if self.two_acc else [(wh, ct) for wh in range(3) for ct in range(NCT)]
                for pos, (which, ct) in enumerate(order):
                        a = V(acc(s, ct), 4)
                        c_in = a
                        if self.two_acc and which != 0:
                            a = V(acc2(s, ct), 4)
                            c_in = "0" if (ks == 0 and which == 1) else a      # the row tile's first correction product starts from zero
                        if is_act:
                            bh, bl = V(hreg(L.src, ct) + 4 * ks, 4), A(lreg(L.src, ct) + 4 * ks, 4)
                        else:
                            k = ks - L.act
                            bh, bl = A(AKEEP + 16 * ct + 4 * k, 4), A(AKEEP + 16 * ct + 8 + 4 * k, 4)
                        if which == 0:
                            e.emit(f16 % (a, A(w, 4), bh, c_in))
                        elif which == 1:
                            e.emit(f16 % (a, A(w + 4, 4), bh, c_in))
                        else:
                            e.emit(f16 % (a, A(w, 4), bl, c_in))
                        self.drain_dma()
                        if pos == len(order) - 1:
                            self.read_frag(fr + NSLOT)          # the slot is free: the fragment eight ahead
                        if gap >= 3 and fillers:
                            remaining = max(n_row - 1 - gap, 0) + 1
                            if tail_of[0] is not None and tail_of[0][0] == L.src and tail_of[0][1] < L.act:
                                first_dep = tail_of[0][1] * 3 * NCT
                                remaining = max(first_dep - 1 - gap, 0) + 1
                            n = (len(fillers) + remaining - 1) // remaining
                            for _ in range(n):
                                if fillers:
                                    run_filler()
                        gap += 1
            if not fillers:
                tail_of[0] = None
        e.emit("s_nop 7")
        e.emit("s_nop 7")
        fillers.extend(epilogue(total_rt - 1))
        flush()
        e.emit("s_waitcnt lgkmcnt(0)")
        e.lds_done = e.lds_issued
        for v in range(self.entered + 1, self.padc):
            self.boundary(v)
        self.drain_dma(99)
        return e.lines


def block_text(lines):
    return "\n".join('        "%s\\n\\t"' % ln for ln in lines)


def emit_pass(name, layers, out, cfg=None):
    gen = PassX3(layers, cfg)
    lines = gen.generate()
    stats = {}
    for ln in lines:
        k = ln.split()[0]
        stats[k] = stats.get(k, 0) + 1
    out.append("// pass %s: %d fragments, %d layers, %d chunks (padded %d); %d instructions; %s" % (
        name, gen.nfrag, len(layers), gen.nchunk, gen.padc, len(lines), ", ".join("%s %d" % kv for kv in sorted(stats.items(), key=lambda kv: -kv[1])[:12])))
    out.append("// keep[ct][0..3] = Ph[0], Ph[1], Pl[0], Pl[1] of column tile ct (fp16 hi / lo B fragments of the point encoding), handed over in")
    out.append("// v60..v91 and moved to a192..a223 by the stream; sigma[ct] (lanes 0..15) comes back in v16, v17.  The stream runs the whole ring")
    out.append("// protocol of one pass (entry, boundaries, walk to the padded end).")
    out.append("constexpr int kX3sFrags = %d, kX3sPadChunks = %d, kX3sChunkBytes = %d;" % (gen.nfrag, gen.padc, CHUNK))
    out.append("template <class Ring>")
    out.append("__device__ __forceinline__ void x3_asm_%s(const Ring& ring, int wave, lds_cptr bias_lane, half8 (&keep)[2][4], float (&sigma)[2]) {" % name)
    out.append("    const unsigned src_lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring.src[0]), src_hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)ring.src[0] >> 32));")
    out.append("    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(size_t)TGTC_LPTR(ring.lds_wave));")

    def rg(base, n):
        return "{v[%d:%d]}" % (base, base + n - 1)
    outs = ['"=&{v%d}"(sigma[%d])' % (OUT + ct, ct) for ct in range(NCT)]
    outs += ['"+%s"(keep[%d][%d])' % (rg(H + 16 * ct + 4 * i, 4), ct, i) for ct in range(NCT) for i in range(4)]
    out.append("    asm volatile(")
    out.append(block_text(lines))
    ins = ['[lane_lo] "v"(ring.lane_lo)', '[lane_hi] "v"(ring.lane_hi)', '[bias_lane] "v"(bias_lane)', '[voff] "v"(ring.voff)',
           '[src_lo] "s"(src_lo)', '[src_hi] "s"(src_hi)', '[ldsw] "s"(ldsw)', '[wave] "s"(wave)']
    clob = ['"memory"', '"scc"', '"m0"', '"s96"', '"s97"']
    pinned = set(range(OUT, OUT + 2)) | set(range(H, H + 32))
    clob += ['"v%d"' % v for v in range(T, LAST_VGPR + 1) if v not in pinned]
    clob += ['"a%d"' % a for a in range(N_AGPR)]
    out.append("        : " + ", ".join(outs))
    out.append("        : " + ", ".join(ins))
    out.append("        : " + ", ".join(clob) + ");")
    out.append("}")
    out.append("")


def set_chunk(chunk):
    """ring of 128 KiB in chunks of `chunk` bytes (16 KiB: 8 slots, 32 KiB: 4 slots, half the boundaries)"""
    global CHUNK, SLOTS, LOOK, GPC, FPC
    CHUNK = chunk
    SLOTS = 131072 // chunk
    LOOK = SLOTS - 1
    GPC = CHUNK // (NWAVES * 1024)
    FPC = CHUNK // FRAG


def main():
    cfg = {}
    for a in sys.argv[1:]:
        k, v = a.split("=")
        cfg[k] = int(v)
    set_chunk(cfg.get("chunk", CHUNK))
    out = ["// GENERATED by tools/gen_x3_asm.py %s-- do not edit; see that file for the design." % ("".join(x + " " for x in sys.argv[1:])), ""]
    emit_pass("nerf_sigma_pass", nerf_sigma_layers(), out, cfg)
    print("\n".join(out))


if __name__ == "__main__":
    main()
