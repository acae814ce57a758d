// TGTC_PREC_FP16_FP6: fp32-equivalent products from ONE fp16 MFMA pass plus two block-scaled fp6 passes.
//
//   W = Wh + Wl (Wh = fp16(W)),  a = Ah + Al (Ah = fp16(a));   W.a ~= Wh.Ah + Wl.Ah + Wh.Al
//   main   Wh.Ah  on v_mfma_f32_16x16x32_f16                     (4 instructions per 128-deep k block)
//   corr1  Wl.Ah  on v_mfma_scale_f32_16x16x128_f8f6f4 (e2m3)    (1 instruction: Wl6 from the stream, Ah6 from registers)
//   corr2  Wh.Al  on the same instruction                        (1 instruction: Wh6 from the stream, Al6 from registers)
// Both correction terms are ~2^-11 of the main term, so their 4-bit (e2m3) operands leave a relative error of
// ~2^-16 per product -- 30 x better than the single fp16 product -- at 6.15 MFMA issue slots per block instead
// of the 12 of TGTC_PREC_FP16X3 (an fp6 16x16x128 costs about what an f16 16x16x32 costs, tools/microbench/mfma_fp6).
// Block scales (E8M0, one per lane = per row/column and 32 k values): weights carry one exponent per output
// row (table in LDS, filled by the packer); activations get theirs from the running maximum of the 32 values a
// lane holds.  Lane maps (probed in tools/microbench/mfma_fp6_check.py): lane l holds row/column l&15 and the 32
// consecutive k = 32(l>>4)+i, six bits each, little endian; code = e2m3(value / scale), RNE, saturating.
// The 32 values a lane holds for four consecutive fp16 k-steps s (element j) are the SAME logical k values
// (i = 8s+j), so Ah6 is one v_cvt_scalef32_pk32_fp6_f16 of registers that already exist.  That instruction
// costs ~100 cycles (tools/microbench/mx_parts), fine once per 128 outputs of an activation block, far too
// slow once per weight group -- so Wh6 is packed on the host and streamed like Wl6.
//
// Positional-encoding / direction k-steps keep the three-product fp16 scheme (their B fragments do not come
// out of an accumulator); NCT = 1 column tile per wave.
#pragma once
#include "mlp_core.h"

// 1: the fp6 correction products accumulate into the row tile's main accumulator (one dependent chain; the two waves of
// a SIMD cover each other's MFMA latency); 0: a separate correction accumulator summed in the epilogue (round 1).
// Measured 101.5 -> 100.2 ms on the headline frame (profiles/r2_kernel_variants.md section 13).
#ifndef TGTC_MX_DEPTH
#define TGTC_MX_DEPTH 1
#endif
#ifndef TGTC_MX_ONE_CHAIN
#define TGTC_MX_ONE_CHAIN 1
#endif

namespace tgtc {

typedef unsigned u6v __attribute__((ext_vector_type(6)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));

// ------------------------------------------------------------------------------------------------ stream layout
// The stream is a sequence of GROUPS, one LDS->register burst each:
//   K group (one 128-deep block of one row tile), 7 KiB:  [Wh k-step 0..3: 4 x 1 KiB][Wl6 dwords 0-3: 1 KiB]
//                                                         [Wl6 dwords 4-5 | Wh6 dwords 0-1: 1 KiB][Wh6 dwords 2-5: 1 KiB],
//                                                         all lane*16: seven ds_read_b128 per group, and the two fp6
//                                                         operands land in twelve consecutive registers without copies
//                                                         (rounds 1-3: 2 x b128 + 2 x b64; the generated streams issue one
//                                                         instruction less per group -- what a lone wave pays for is issue)
//   P group (the PE / direction k-steps of one row tile), npe x 2 KiB: [hi, lo] per k-step, lane*16
// Groups never straddle the end of the 128 KiB ring (padding inserted), they may straddle 16 KiB chunks.
struct MxShape {
    int rt, nkb, npe;
};
constexpr int kMxKGroupBytes = 7168;
constexpr int kMxMaxGroups = 512;

struct MxTable {
    int n = 0;
    int off[kMxMaxGroups] = {};
    short npe[kMxMaxGroups] = {};  // 0: K group
    int first[16] = {};            // first group of layer l; first[nl] = n
    int bytes = 0;                 // stream length, chunk padded
};

template <int NL>
constexpr MxTable mx_make_table(const MxShape (&s)[NL], int ring_bytes) {
    MxTable t;
    int off = 0;
    for (int l = 0; l < NL; ++l) {
        t.first[l] = t.n;
        for (int rt = 0; rt < s[l].rt; ++rt) {
            const int ng = s[l].nkb + (s[l].npe ? 1 : 0);
            for (int q = 0; q < ng; ++q) {
                const int npe = q < s[l].nkb ? 0 : s[l].npe;
                const int size = npe ? npe * 2048 : kMxKGroupBytes;
                if (off / ring_bytes != (off + size - 1) / ring_bytes) off = (off / ring_bytes + 1) * ring_bytes;
                t.off[t.n] = off, t.npe[t.n] = (short)npe, ++t.n;
                off += size;
            }
        }
    }
    t.first[NL] = t.n;
    t.bytes = (off + kChunkBytes - 1) / kChunkBytes * kChunkBytes;
    return t;
}

//   layer      L0  L1  L2  L3  L4  L5  L6  L7  SIG REMAP C0  C1      (mlp_layouts.h)
constexpr MxShape kNerfMxShape[12] = {{16, 0, 2}, {16, 2, 0}, {16, 2, 0}, {16, 2, 0}, {16, 2, 0}, {16, 2, 2},
                                      {16, 2, 0}, {16, 2, 0}, {1, 2, 0},  {16, 2, 0}, {8, 2, 1},  {1, 1, 0}};
inline constexpr MxTable kNerfMxTable = mx_make_table(kNerfMxShape, kRingBytes);
constexpr int kNerfMxScaleOff = 10240;  // byte offset of the row-exponent table inside the bias region (u16 per row)

// e2m3: 1 sign, 2 exponent (bias 1), 3 mantissa bits; codes are monotone in magnitude
__host__ __device__ inline float e2m3_value(int code) {
    const int e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m * 0.125f : (1.0f + m * 0.125f) * (float)(1 << (e - 1));
    return (code & 32) ? -v : v;
}

// ------------------------------------------------------------------------------------------------ device side
// LDS -> register staging of one group at a time.  The register budget (8 waves x 256 VGPRs, two layers of
// activations resident) has no room for a second group buffer, so the NEXT group's units are read into the
// registers of the current group as soon as the MFMA that consumed each of them has issued (refill<>), and the
// single lgkmcnt(0) the compiler emits lands at the start of the next group (acquire<>).
template <class C, class Map, const MxTable& T, bool PERSIST = false, int DEPTH_ = TGTC_MX_DEPTH>
struct MxReader {
    static constexpr int DEPTH = DEPTH_;   // groups staged ahead of the one being multiplied (1 or 2 register buffers)
    using Ring = WeightStream<C, Map, PERSIST, true>;   // counted LDS waits need the DMA hidden in asm (mlp_core.h)
    Ring ring;
    __device__ __forceinline__ static const char* lane_src(const char* stream, int wave, int lane) {
        return Ring::lane_src(stream, wave, lane);
    }
    lds_cptr b8_lo, b8_hi;  // ring + lane*8, lower / upper 64 KiB
    half8 ub[DEPTH][4];     // [group % DEPTH] units 0..3: fp16 fragments (K group: Wh k-steps; P group: hi/lo pairs)
    u6v wb[DEPTH][2];       // K group: Wl6 (units 4 = dwords 0-3, 6 = dwords 4-5), Wh6 (units 5, 7)

    __device__ __forceinline__ void init(const char* const (&streams)[Map::NSEG], char* smem, int wave, int lane) {
        ring.init(streams, smem, wave, lane);
        b8_lo = opaque((lds_cptr)smem + lane * 8);
        b8_hi = opaque((lds_cptr)smem + (C::RING_BYTES > 65536 ? 65536 : 0) + lane * 8);
    }
    // Re-derive every per-lane address of the reader from a freshly read lane id (around the generated asm blocks of
    // mx_asm_nerf.inc: values that are live across such a block are spilled by hipcc -- the block clobbers v40..v251 -- and
    // a scratch reload waits vmcnt(0), i.e. for the ring's whole look-ahead; values that die in front of it are not)
    __device__ __forceinline__ void relane(char* smem, int wave) {
        const int lane = fresh_lane_id();
        ring.voff = wave * (C::GPC * 1024) + lane * 16;
        ring.lane_lo = opaque((lds_cptr)smem + lane * 16);
        ring.lane_hi = opaque((lds_cptr)smem + (C::RING_BYTES > 65536 ? 65536 : 0) + lane * 16);
        b8_lo = opaque((lds_cptr)smem + lane * 8);
        b8_hi = opaque((lds_cptr)smem + (C::RING_BYTES > 65536 ? 65536 : 0) + lane * 8);
    }
    template <int OFF>
    __device__ __forceinline__ half8 read16() const {
        constexpr int o = OFF % C::RING_BYTES;
        typedef __attribute__((address_space(3))) const half8* p_t;
        if constexpr (o < 65536) return *(p_t)(ring.lane_lo + o);
        else return *(p_t)(ring.lane_hi + (o - 65536));
    }
    template <int OFF>
    __device__ __forceinline__ u2v read8() const {
        constexpr int o = OFF % C::RING_BYTES;
        typedef __attribute__((address_space(3))) const u2v* p_t;
        if constexpr (o < 65536) return *(p_t)(b8_lo + o);
        else return *(p_t)(b8_hi + (o - 65536));
    }
    static constexpr int units(int qi) { return T.npe[qi] ? 2 * T.npe[qi] : 8; }
    static constexpr int bytes(int qi) { return T.npe[qi] ? 2048 * T.npe[qi] : kMxKGroupBytes; }
    static constexpr int chunk_hi(int qi) { return (T.off[qi] + bytes(qi) - 1) / kChunkBytes; }
    // read unit J of group Q (no-op past the end of the stream / of the group)
    template <int Q, int NQ, int J>
    __device__ __forceinline__ void refill() {
        if constexpr (Q < NQ) {
            if constexpr (J < units(Q)) {
                if constexpr (J < 4) {
                    ub[Q % DEPTH][J] = read16<T.off[Q] + 1024 * J>();
                } else if constexpr (J == 4) {          // piece A: Wl6 dwords 0-3 (its registers are free once C1 has issued)
                    const u4v t = __builtin_bit_cast(u4v, read16<T.off[Q] + 4096>());
                    u6v& w = wb[Q % DEPTH][0];
                    w[0] = t[0], w[1] = t[1], w[2] = t[2], w[3] = t[3];
                } else if constexpr (J == 5) {          // piece C: Wh6 dwords 2-5 (free once C2 has issued)
                    const u4v t = __builtin_bit_cast(u4v, read16<T.off[Q] + 6144>());
                    u6v& w = wb[Q % DEPTH][1];
                    w[2] = t[0], w[3] = t[1], w[4] = t[2], w[5] = t[3];
                } else if constexpr (J == 7) {          // piece B: Wl6 dwords 4-5 | Wh6 dwords 0-1 -- needs BOTH operands released,
                    const u4v t = __builtin_bit_cast(u4v, read16<T.off[Q] + 5120>());   // so it rides with unit 5, behind C2
                    wb[Q % DEPTH][0][4] = t[0], wb[Q % DEPTH][0][5] = t[1];
                    wb[Q % DEPTH][1][0] = t[2], wb[Q % DEPTH][1][1] = t[3];
                }                                       // (J == 6: nothing -- seven LDS reads per K group, round 4)
            }
        }
    }
    // units [J0, 8) of group Q
    template <int Q, int NQ, int J0>
    __device__ __forceinline__ void refill_from() {
        static_for<8 - J0>([&](auto j) { refill<Q, NQ, J0 + decltype(j)::value>(); });
    }
    // ring prologue was issued by the caller (ring.prologue()); wait for chunks 0,1 and read group Q0
    template <int Q0, int NQ>
    __device__ __forceinline__ void start() {
        static_assert(chunk_hi(Q0 + DEPTH - 1) <= 1, "the first groups must lie in chunks 0..1");
        ring.start_ring();
        static_for<DEPTH>([&](auto d) { refill_from<Q0 + decltype(d)::value, NQ, 0>(); });
        __builtin_amdgcn_sched_barrier(0);
    }
    // PERSIST (fused ray kernel): enter the stream ring.next points to and read group Q0; finish<NQ>() walks the ring
    // to the end of the padded pass (WeightStream, PERSIST): acquire<> has entered chunks up to chunk_hi(NQ-1) - 1.
    template <int Q0, int NQ>
    __device__ __forceinline__ void enter() {
        static_assert(chunk_hi(Q0 + DEPTH - 1) <= 1, "the first groups must lie in chunks 0..1");
        ring.enter_ring();
        static_for<DEPTH>([&](auto d) { refill_from<Q0 + decltype(d)::value, NQ, 0>(); });
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int NQ>
    __device__ __forceinline__ void finish() const {
        ring.template finish<(chunk_hi(NQ - 1) > 1 ? chunk_hi(NQ - 1) : 1)>();
    }
    // Entering group Q: acquire the chunks group Q+1 touches.  A boundary re-fills the slot of chunk c-1, and group Q
    // itself may straddle chunks c-1 | c with its reads still in flight (they were issued while group Q-1 ran), so
    // those reads are retired first: the empty asm "uses" make hipcc place its counted wait for them in front of the
    // barrier (volatile asm statements and the barrier keep their order).  With groups of at most half a chunk the
    // case cannot arise (group Q+1 would have to cover a whole chunk), so today this compiles to nothing; it guards
    // the invariant should the group or chunk size change.
    // the last chunk that must have been entered so that group q + DEPTH (the one refilled while group q runs) is readable
    static constexpr int last_needed(int q, int nq) {
        const int g = q + DEPTH < nq - 1 ? q + DEPTH : nq - 1;
        return chunk_hi(g) - 1;
    }
    // one counted wait for the whole staged group G (the empty asm "uses" all its units); pays when the group was read two
    // groups ago (DEPTH 2): the per-MFMA waits hipcc would otherwise emit disappear
    template <int G, int NQ>
    __device__ __forceinline__ void touch() {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass instantiates this body through dense_mx's generic lambdas and has no "v" registers)
        if constexpr (G < NQ) {
            static_for<(units(G) < 4 ? units(G) : 4)>([&](auto j) { asm volatile("" ::"v"(ub[G % DEPTH][decltype(j)::value])); });
            if constexpr (units(G) == 8) asm volatile("" ::"v"(wb[G % DEPTH][0]), "v"(wb[G % DEPTH][1]));
        }
#endif
    }
    // make the reads of staged group G complete if its bytes reach back into chunk HI-1 (whose slot is about to be re-filled)
    template <int G, int HI, int NQ>
    __device__ __forceinline__ void retire() {
        if constexpr (G < NQ) {
            if constexpr (T.off[G] / kChunkBytes < HI) {
                static_for<(units(G) < 4 ? units(G) : 4)>([&](auto j) { asm volatile("" ::"v"(ub[G % DEPTH][decltype(j)::value])); });
                if constexpr (units(G) == 8) asm volatile("" ::"v"(wb[G % DEPTH][0]), "v"(wb[G % DEPTH][1]));
            }
        }
    }
    // the boundaries waves 0..3 run on entering group QQ, run at the start of group Q (QQ = Q, or Q + STAG for waves 4..7)
    template <int Q, int QQ, int NQ>
    __device__ __forceinline__ void acquire_at() {
        constexpr int prev = Q == 0 ? 0 : last_needed(QQ - 1, NQ);
        constexpr int lo = prev + 1 > 1 ? prev + 1 : 1, hi = last_needed(QQ, NQ);
        if constexpr (hi >= lo) {
            // boundary<hi> re-fills the slot of chunk hi-1 (hi-2 with a stagger): groups already staged in registers whose
            // bytes reach back into it must have their reads retired first
            constexpr int refilled = hi - (Ring::STAG > 0 ? 1 : 0);
            retire<Q, refilled, NQ>();
            if constexpr (DEPTH > 1) retire<Q + 1, refilled, NQ>();
            static_for<hi - lo + 1>([&](auto i) { ring.template boundary<lo + decltype(i)::value>(); });
        }
    }
    template <int Q, int QQ, int NQ>
    static constexpr bool acquires_at() {
        constexpr int prev = Q == 0 ? 0 : last_needed(QQ - 1, NQ);
        return last_needed(QQ, NQ) >= (prev + 1 > 1 ? prev + 1 : 1);
    }
    template <int Q, int NQ>
    __device__ __forceinline__ void acquire() {
        if constexpr (Ring::STAG == 0) {
            acquire_at<Q, Q, NQ>();
        } else if constexpr (acquires_at<Q, Q, NQ>() || acquires_at<Q, Q + Ring::STAG, NQ>()) {
            if (ring.late) acquire_at<Q, Q + Ring::STAG, NQ>();
            else acquire_at<Q, Q, NQ>();
        }
    }
};

__device__ __forceinline__ i8v widen6(u6v r) {
    return i8v{(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0};
}
// acc += (A6 * 2^(sa.byte[OPA]-127)) . (B6 * 2^(sb.byte[OPB]-127)),  16x16x128, both operands e2m3
template <int OPA, int OPB>
__device__ __forceinline__ float4v mfma_fp6(u6v a, u6v b, float4v c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(widen6(a), widen6(b), c, 2, 2, OPA, sa, OPB, sb);
}
__device__ __forceinline__ u6v cvt_fp6(half8 a, half8 b, half8 c, half8 d, float scale) {
    half32 v = __builtin_shufflevector(__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15),
                                       __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15),
                                       0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23,
                                       24, 25, 26, 27, 28, 29, 30, 31);
    return __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
}

// ReLU + hi/lo split of an accumulator pair without fp32 arithmetic next to the MFMAs (which is what costs there,
// profiles/r1_kernel_variants.md): two integer max, one packed convert, and the lo halves straight from
// v_fma_mixlo/mixhi_f16 (v * 1.0 - h in fp32, rounded once to fp16).  lo is NOT pre-scaled: it may be an fp16
// subnormal (|lo| <= 2^-12 h), which v_cvt_scalef32_pk32_fp6_f16 takes like any other value, and the block scale of
// the lo operand carries the 2^-11 instead.
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, b)));
}
// E8M0 byte of the block scale from the packed running maximum of non-negative fp16 values: block max in
// [2^E, 2^(E+1)) -> scale 2^(E-1): codes in [2,4), no saturation.  fp16 exponent field e16 = E+15, byte = E-1+127.
__device__ __forceinline__ int block_exp_byte(unsigned mxk) {
    const unsigned m = max(mxk & 0xffffu, mxk >> 16);
    return (int)(m >> 10) + 111;
}

// Activations of one layer as the next layer's B operands: h[4*NKB] fp16 k-steps, h6 / l6 fp6 blocks,
// sc = E8M0 bytes (byte 0: h6, byte 1: l6).  l16 is the staging area of the block being produced.
template <int NKB>
struct MxAct {
    half8 h[4 * NKB];
    u6v h6[NKB], l6[NKB];
    int sc[NKB];
    unsigned mxk;  // packed running maximum of the block being produced
};

// ReLU + split of one HALF of row tile RT's accumulator; closes k block RT/8 when its last values arrive.
template <int RT, int HALF, int NKB>
__device__ __forceinline__ void mx_store_act(const float4v& acc, MxAct<NKB>& y, half8 (&l16)[4]) {
    constexpr int ks = RT / 2, e0 = (RT & 1) * 4 + 2 * HALF;
    unsigned hpk, lpk;
    split_pair(acc[2 * HALF], acc[2 * HALF + 1], hpk, lpk);
    set_pair(y.h[ks], e0, hpk);
    set_pair(l16[ks & 3], e0, lpk);
    y.mxk = ((RT & 7) == 0 && HALF == 0) ? hpk : pk_max_u16(y.mxk, hpk);
    if constexpr ((RT & 7) == 7 && HALF == 1) {
        constexpr int kb = RT / 8;
        const int byte_h = block_exp_byte(y.mxk);
        y.h6[kb] = cvt_fp6(y.h[4 * kb], y.h[4 * kb + 1], y.h[4 * kb + 2], y.h[4 * kb + 3], __builtin_bit_cast(float, byte_h << 23));
        // |lo| <= half an ulp of the block maximum = 2^(E-11): with scale 2^(E-13) its codes reach 4 of e2m3's 7.5, one bit
        // finer than the 2^(E-12) of round 2 and still free of saturation
        y.l6[kb] = cvt_fp6(l16[0], l16[1], l16[2], l16[3], __builtin_bit_cast(float, (byte_h - 12) << 23));
        y.sc[kb] = byte_h | ((byte_h - 12) << 8);
    }
}

// One dense layer.  Groups Q0 + rt*(NKB + (NPE>0)) + i.  X: activation operands (NKB blocks), Ph/Pl: the NPE
// fp16 hi/lo k-steps (encodings).  epi(ic<rt>, ic<half>, acc) as in dense_layer, NCT = 1.
// rs_lane: LDS address of the row-exponent table + 2*(lane&15) (u16: byte 0 = Wh6 exponent, byte 1 = Wl6's).
// One accumulator chain per row tile (TGTC_MX_ONE_CHAIN; round 1 kept the fp6 products in a second accumulator and
// summed the two in the epilogue: four moves and four adds per row tile more, no faster).  The instruction order is pinned
// step by step (sched_barrier): one MFMA, the refill reads of the unit it consumed, a slice of VALU work.
struct MxNoTrace {
    template <int I>
    __device__ __forceinline__ void operator()(ic<I>) const {}
};
template <class C, int Q0, int NQ, int RT, int NKB, int NPE, int BIAS0, class Reader, class Epi, class Trace = MxNoTrace>
__device__ __forceinline__ void dense_mx(Reader& rd, lds_cptr bias_lane, lds_cptr rs_lane, const MxAct<(NKB ? NKB : 1)>& X,
                                         const half8 (&Ph)[NPE ? NPE : 1], const half8 (&Pl)[NPE ? NPE : 1], Epi&& epi,
                                         Trace trace = Trace{}) {
    constexpr int GPR = NKB + (NPE ? 1 : 0);
    typedef __attribute__((address_space(3))) const float4v* lds_f4;
    typedef __attribute__((address_space(3))) const unsigned short* lds_u16;
    float4v accm[2], accc[2];
    constexpr bool ONE = TGTC_MX_ONE_CHAIN && NKB > 0;
    int rs[2] = {0, 0};
    accm[0] = *(lds_f4)(bias_lane + BIAS0 * 4);
    if constexpr (NKB > 0) rs[0] = *(lds_u16)(rs_lane + BIAS0 * 2);
    auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
    static_for<RT>([&](auto rt_) {
        constexpr int rt = decltype(rt_)::value;
        constexpr int cur = rt & 1;
        if constexpr (NKB > 0 && !ONE) accc[cur] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
        static_for<GPR>([&](auto gi_) {
            constexpr int gi = decltype(gi_)::value;
            constexpr int Q = Q0 + rt * GPR + gi;
            // the deferred epilogue of the previous row tile (one half per group), then the next row tile's bias
            // straight into the accumulator that epilogue has just released
            auto deferred = [&] {
                if constexpr (rt > 0) {
                    float4v sum = accm[cur ^ 1];
                    if constexpr (NKB > 0 && !ONE) sum += accc[cur ^ 1];
                    if constexpr (GPR == 1) {
                        epi(ic<rt - 1>{}, ic<0>{}, sum);
                        epi(ic<rt - 1>{}, ic<1>{}, sum);
                    } else if constexpr (gi < 2) {
                        epi(ic<rt - 1>{}, ic<gi>{}, sum);
                    }
                }
                if constexpr (gi == (GPR == 1 ? 0 : 1) && rt + 1 < RT) {
                    accm[cur ^ 1] = *(lds_f4)(bias_lane + (BIAS0 + 16 * (rt + 1)) * 4);
                    if constexpr (NKB > 0) rs[cur ^ 1] = *(lds_u16)(rs_lane + (BIAS0 + 16 * (rt + 1)) * 2);
                }
            };
            rd.template acquire<Q, NQ>();
            trace(ic<rt * GPR + gi>{});
            constexpr int D = Reader::DEPTH;
            if constexpr (D > 1) rd.template touch<Q, NQ>();
            half8 (&U)[4] = rd.ub[Q % D];
            u6v (&W)[2] = rd.wb[Q % D];
            if constexpr (gi < NKB) {
                constexpr int kb = gi;
                accm[cur] = mfma16(U[0], X.h[4 * kb + 0], accm[cur]);
                rd.template refill<Q + D, NQ, 0>();
                fence();
                if constexpr (ONE) accm[cur] = mfma_fp6<1, 0>(W[0], X.h6[kb], accm[cur], rs[cur], X.sc[kb]);
                else accc[cur] = mfma_fp6<1, 0>(W[0], X.h6[kb], accc[cur], rs[cur], X.sc[kb]);
                rd.template refill<Q + D, NQ, 4>();
                rd.template refill<Q + D, NQ, 6>();
                fence();
                accm[cur] = mfma16(U[1], X.h[4 * kb + 1], accm[cur]);
                rd.template refill<Q + D, NQ, 1>();
                deferred();  // early in the group: the bias / row-exponent reads it ends with are needed at the next row tile's start
                fence();
                accm[cur] = mfma16(U[2], X.h[4 * kb + 2], accm[cur]);
                rd.template refill<Q + D, NQ, 2>();
                fence();
                if constexpr (ONE) accm[cur] = mfma_fp6<0, 1>(W[1], X.l6[kb], accm[cur], rs[cur], X.sc[kb]);
                else accc[cur] = mfma_fp6<0, 1>(W[1], X.l6[kb], accc[cur], rs[cur], X.sc[kb]);
                rd.template refill<Q + D, NQ, 5>();
                rd.template refill<Q + D, NQ, 7>();
                fence();
                accm[cur] = mfma16(U[3], X.h[4 * kb + 3], accm[cur]);
                rd.template refill<Q + D, NQ, 3>();
                fence();
            } else {
                static_for<NPE>([&](auto k_) {
                    constexpr int k = decltype(k_)::value;
                    accm[cur] = mfma16(U[2 * k], Ph[k], accm[cur]);
                    if constexpr (NKB > 0 && !ONE) accc[cur] = mfma16(U[2 * k + 1], Ph[k], accc[cur]);
                    else accm[cur] = mfma16(U[2 * k + 1], Ph[k], accm[cur]);
                    accm[cur] = mfma16(U[2 * k], Pl[k], accm[cur]);
                    rd.template refill<Q + D, NQ, 2 * k>();
                    rd.template refill<Q + D, NQ, 2 * k + 1>();
                    fence();
                });
                rd.template refill_from<Q + D, NQ, 2 * NPE>();
                deferred();
                fence();
            }
        });
    });
    // hipcc's hazard model gives VALU readers of a v_mfma_scale_f32_16x16x128_f8f6f4 result the wait states of a
    // 4-pass MFMA; on gfx950 hardware that is not enough (stale accumulators were read right behind the last fp6
    // MFMA of a layer).  The deferred epilogues are >= 6 MFMAs behind; only this final one needs the explicit drain.
    float4v sum = accm[(RT - 1) & 1];
    if constexpr (ONE) {
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sum));
    } else if constexpr (NKB > 0) {
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(accc[(RT - 1) & 1]));
        sum += accc[(RT - 1) & 1];
    }
    epi(ic<RT - 1>{}, ic<0>{}, sum);
    epi(ic<RT - 1>{}, ic<1>{}, sum);
}

}  // namespace tgtc
