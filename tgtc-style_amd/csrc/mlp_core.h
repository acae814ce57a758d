// Building blocks of the fused "positional encoding + MLP" kernels (gfx950 / CDNA4 only).
//
// Formulation: every layer is evaluated TRANSPOSED, H_out^T[features x samples] =
// W[features x K] * H_in^T[K x samples], on v_mfma_f32_16x16x32_f16:
//   * A operand  = a 16(out-feature) x 32(k) weight fragment, read from an LDS ring that is filled by
//                  LDS-DMA (global_load_lds_dwordx4) from a pre-packed, fragment-ordered stream in HBM/L2;
//   * B operand  = activations, which NEVER leave registers: the 16x16 fp32 accumulator tile of layer l
//                  (feature rows 4*(lane>>4)+r, sample column lane&15) is, after ReLU and conversion to
//                  fp16, exactly the B fragment layout of layer l+1 for a k-permutation that is folded
//                  into the weight packing (two 16-row tiles form one 32-deep k-step);
//   * each wavefront owns NCT column tiles of 16 samples and all output features; the 4 waves of a
//     workgroup share one weight ring, so the weights cross L2->LDS once per 4*NCT*16 samples.
// Precision: TGTC_PREC_FP16 = one MFMA per product; TGTC_PREC_FP16X3 = weights and activations split
// into fp16 hi+lo, three MFMAs (hi*hi + lo*hi + hi*lo) per product, fp32 accumulate: fp32-equivalent.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace tgtc {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define TGTC_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TGTC_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int I>
using ic = std::integral_constant<int, I>;

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(ic<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// LDS-DMA: 16 bytes per lane from global memory to  lds_dst (wave-uniform) + lane*16.
// With TGTC_ASM_DMA the instruction is emitted through inline asm: hipcc's waitcnt insertion treats a
// global_load_lds it can see as an LDS event of a second kind, and from then on every wait for a ds_read is
// `lgkmcnt(0)` instead of a counted wait (each LDS read then waits for ALL outstanding reads).  Hidden in asm, the
// ds_reads get exact counted waits; the DMA's own completion is tracked by hand anyway (wait_vmcnt + s_barrier).
// M0 carries the LDS destination; nothing else in these kernels uses M0.
// The fp16 / fp16x3 layer loop was tuned around the builtin (bursts that expect lgkmcnt(0)): hidden in asm it runs
// 2.6 % slower (profiles/r2_kernel_variants.md), so the choice belongs to the stream, not to the translation unit.
#ifdef TGTC_ASM_DMA
constexpr bool kAsmDmaDefault = true;
#else
constexpr bool kAsmDmaDefault = false;
#endif
template <bool ASM>
__device__ __forceinline__ void lds_dma16_t(const char* gsrc, char* lds_dst) {
    if constexpr (ASM) {
        const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)TGTC_LPTR(lds_dst));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(l) : "memory");
    } else {
        __builtin_amdgcn_global_load_lds(TGTC_GPTR(gsrc), TGTC_LPTR(lds_dst), 16, 0, 0);
    }
}
__device__ __forceinline__ void lds_dma16(const char* gsrc, char* lds_dst) { lds_dma16_t<kAsmDmaDefault>(gsrc, lds_dst); }
// The same with the address split as the instruction wants it: a wave-uniform 64-bit base in SGPRs, a 32-bit per-lane
// offset and a 13-bit immediate -- no 64-bit VALU add per instruction (the builtin never selects this form: it always
// adds up a per-lane 64-bit address with v_lshl_add_u64).  The immediate is added to BOTH addresses -- the global one and
// the LDS destination (M0 + offset + lane*16) -- so `lds_dst` is the destination of IMM = 0.
template <int IMM>
__device__ __forceinline__ void lds_dma16_s(const char* sbase, unsigned voff, char* lds_dst) {
    static_assert(IMM >= 0 && IMM < 4096, "global instruction offset field");
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)TGTC_LPTR(lds_dst));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(l), "n"(IMM) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------
// Geometry shared by the packer (host) and the kernels (device).
constexpr int kFragHalves = 64 * 8;       // one MFMA A fragment: 64 lanes x 8 halves = 1 KiB
#ifndef TGTC_CHUNK_BYTES
#define TGTC_CHUNK_BYTES 16384
#endif
constexpr int kChunkBytes = TGTC_CHUNK_BYTES;           // ring granule
// Timing experiments (development builds only, results are garbage): bit 0 no LDS-DMA after the prologue, bit 1 no ring
// barrier, bit 2 no LDS->register fragment reads after the first group, bit 3 epilogues reduced to one instruction.
#ifndef TGTC_ABL
#define TGTC_ABL 0
#endif
constexpr int kAbl = TGTC_ABL;
constexpr int kRingSlots = 131072 / kChunkBytes;         // 128 KiB ring
constexpr int kRingBytes = kChunkBytes * kRingSlots;
constexpr int kPrefetchDepth = kRingSlots - 1;

template <int NWAVES_, int NCT_, bool SPLIT_, int G_ = (SPLIT_ ? 4 : 8), int SLOTS_ = kRingSlots, int WG_PER_CU_ = 1,
          bool PARK_ = false>
struct MlpCfg {
    // PARK: one wave per SIMD (512 registers per lane).  The layer being produced is parked in the accumulator
    // half of the register file (v_accvgpr_write) and copied back at the layer boundary, so that every MFMA
    // operand is an architectural VGPR (an MFMA whose B operand is an AGPR issues ~25 % slower).
    static constexpr bool PARK = PARK_;
    static constexpr int SLOTS = SLOTS_;                   // 16 KiB ring slots of this workgroup
    static constexpr int RING_BYTES = SLOTS_ * kChunkBytes;
    static constexpr int WG_PER_CU = WG_PER_CU_;           // co-resident workgroups the LDS budget is sized for
    static constexpr int NWAVES = NWAVES_;
    static constexpr int NCT = NCT_;                       // 16-sample column tiles per wave
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int G = G_;                           // fragments per LDS->register staging group
    static constexpr int FRAG_BYTES = SPLIT ? 2048 : 1024;  // hi (+ lo) fragment
    static constexpr int FPC = kChunkBytes / FRAG_BYTES;    // fragments per chunk
    static constexpr int GPC = kChunkBytes / (NWAVES * 1024);  // LDS-DMA instructions per wave per chunk
    static constexpr int SAMPLES_PER_WAVE = NCT * 16;
    static constexpr int SAMPLES_PER_WG = NWAVES * SAMPLES_PER_WAVE;
};

// ------------------------------------------------------------------------------------------------
// Weight stream: NCHUNK chunks of the packed fragment stream flow through kRingSlots LDS slots
// (LDS-DMA, no VGPRs), and from there through a small register queue that runs PF fragments ahead of
// the MFMAs that consume them.
//
// Ring protocol.  prologue() issues chunks 0..7.  start() waits (counted vmcnt) for this wave's pieces
// of chunks 0 and 1 and joins the workgroup barrier.  On entering chunk CH >= 1 (boundary<CH>) a wave
// waits for its pieces of chunk CH+1, joins the barrier -- which makes chunk CH+1 visible to every
// wave and proves every wave is done with chunk CH-1 -- and then re-fills chunk CH-1's slot with chunk
// CH+7.  Acquiring one chunk AHEAD lets the register queue read across chunk boundaries, so the MFMA
// pipe never drains at a boundary.
typedef __attribute__((address_space(3))) const char* lds_cptr;

// The lane id, re-read from the hardware where it is called: `threadIdx.x & 63` (and any pure function of it) is computed
// once per kernel by hipcc and kept -- or spilled -- from there on; a volatile asm is evaluated in place.
__device__ __forceinline__ int fresh_lane_id() {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

__device__ __forceinline__ lds_cptr opaque(lds_cptr p) {
    // hide the constant relation between LDS base registers from the optimiser, so that every
    // fragment read is `ds_read_b128 v, base offset:imm16` (no per-read address arithmetic)
    unsigned a = (unsigned)(size_t)p;
    asm volatile("" : "+v"(a));
    return (lds_cptr)(size_t)a;
}

// Map: where the fragment stream comes from.  Map::NFRAG fragments in total, laid out as Map::NSEG
// chunk-aligned segments; segment i covers chunks [Map::chunk0(i), Map::chunk0(i+1)) and is read from
// its own device pointer (so one fused kernel can walk the concat-MLP, NeRF and style-MLP streams of
// three separately packed handles back to back).
template <int NFRAG_>
struct SingleStreamMap {
    static constexpr int NFRAG = NFRAG_;
    static constexpr int NSEG = 1;
    static constexpr int chunk0(int i) { return i == 0 ? 0 : (1 << 30); }
};

//
// PERSIST (fused ray kernel, render_fused.hip): a wave walks stream after stream without ever draining the ring.
// A pass's stream is rounded up to PADC = a multiple of SLOTS chunks, so that the next pass starts in slot 0 again
// and every LDS offset stays a compile-time constant; the PADC - NCHUNK dummy chunks are fetched like real ones
// (they are the bytes behind the logical stream, inside the handle's allocation) and never read.  The protocol is
// then uniform: entering virtual chunk v (a consumed chunk, a dummy at the end of the pass -- finish() -- or chunk
// 0 of the next pass -- enter()) waits until at most SLOTS-3 chunks are in flight (chunks v, v+1 have landed),
// joins the workgroup barrier and issues chunk v+SLOTS-1, which for v+SLOTS-1 >= PADC is chunk v+SLOTS-1-PADC of
// the stream `next` points to.  Other vector-memory operations of the wave (ray loads, pixel stores) only make
// the counted waits conservative: vmcnt retires in issue order.
//
// STAGGER (PERSIST only, -DTGTC_STAGGER=S, S > 0): the two waves of a SIMD (waves w and w + 4 of the workgroup) run the
// same program and the ring barrier keeps them in phase -- both reach their LDS bursts, their epilogue VALU work and
// their MFMA runs together (MI355X_MICROARCH.md, "Two waves per SIMD", item 9; profiles/r1_kernel_variants.md: the
// barrier costs by keeping the SIMD-sharing waves in phase).  With a stagger, waves 4..7 ("late") execute every ring
// boundary S register groups EARLIER in their program than waves 0..3: all eight waves still meet at the same barriers,
// so behind each barrier the late waves are S groups behind their partners -- locked in anti-phase instead of in phase.
// A late wave is still reading chunk v-1 when it joins the barrier "entering v", so the slot that barrier re-fills is
// the one of chunk v-2: the look-ahead is SLOTS-2 chunks instead of SLOTS-1.
#ifndef TGTC_STAGGER
#define TGTC_STAGGER 0
#endif
template <class C, class Map, bool PERSIST = false, bool ASM_DMA = kAsmDmaDefault>
struct WeightStream {
    static constexpr bool IS_PERSIST = PERSIST;
    static constexpr int NFRAG = Map::NFRAG;
    static constexpr int NCHUNK = (NFRAG + C::FPC - 1) / C::FPC;
    static constexpr int PADC = PERSIST ? (NCHUNK + C::SLOTS - 1) / C::SLOTS * C::SLOTS : NCHUNK;
    static constexpr int STAG = PERSIST ? TGTC_STAGGER : 0;           // register groups by which waves 4..7 run behind
    static constexpr int LOOK = C::SLOTS - 1 - (STAG > 0 ? 1 : 0);    // PERSIST: chunks issued ahead of the one being entered
    bool late = false;                                                 // wave-uniform: wave >= 4 (STAG > 0 only)
    // (a persistent stream of several segments -- the stylised ray kernel's concat | NeRF trunk | style -- re-enters its own
    // first segment or hands over to another stream through `next`; its dummy chunks are the bytes behind the LAST segment)
    // LDS -> register staging in bursts of G fragments, double buffered: group g+1 is read while the
    // MFMAs of group g run.  (hipcc only ever emits `s_waitcnt lgkmcnt(0)` here, never a counted wait,
    // so each wait must find every outstanding read already old: one burst per group, issued right
    // after the previous group's wait.)
    static constexpr int G = C::G;
    static_assert(C::FPC % G == 0 && 2 * G <= C::FPC, "groups must tile a chunk and stay within one chunk of look-ahead");
    // LDS reads this window issues (fragments 2k, 2k+1 of the NEXT group while 2k < G), hi (+ lo) each
    static constexpr int reads_in_window(int f) { return (2 * (f % G) < G ? 2 : 0) * (C::SPLIT ? 2 : 1); }

    // stream pointers: per lane (segment stream + wave*GPC*1024 + lane*16) for the builtin DMA; with ASM_DMA the
    // wave-uniform stream base, the per-lane part being `voff` (lds_dma16_s)
    const char* src[Map::NSEG];
    const char* next;            // PERSIST: the same for the stream of the next pass
    unsigned voff;               // wave*GPC*1024 + lane*16
    char* lds_wave;              // wave-uniform: ring + wave*GPC*1024
    lds_cptr lane_lo;  // ring + lane*16            (ring bytes [0, 64K))
    lds_cptr lane_hi;  // ring + 65536 + lane*16    (ring bytes [64K, 128K))
    half8 qh[2][G], ql[2][G];

    __device__ __forceinline__ void init(const char* const (&streams)[Map::NSEG], char* smem, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < Map::NSEG; ++i) src[i] = lane_src(streams[i], wave, lane);
        voff = wave * (C::GPC * 1024) + lane * 16;
        late = STAG > 0 && wave >= C::NWAVES / 2;
        lds_wave = smem + wave * (C::GPC * 1024);
        lane_lo = opaque((lds_cptr)smem + lane * 16);
        lane_hi = opaque((lds_cptr)smem + (C::RING_BYTES > 65536 ? 65536 : 0) + lane * 16);
    }
    static constexpr int seg_of(int ch) {
        int s = 0;
        for (int i = 1; i < Map::NSEG; ++i)
            if (ch >= Map::chunk0(i)) s = i;
        return s;
    }
    template <size_t OFF, int SLOT>
    __device__ __forceinline__ void issue_chunk(const char* base) const {
        // launder the base: inside a persistent tile loop every chunk address is loop invariant, and hipcc would
        // otherwise hoist hundreds of 64-bit addresses into the pre-header and spill them
        if constexpr (ASM_DMA) {
            asm volatile("" : "+s"(base));
            const char* b = base + OFF;
            static_for<C::GPC>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                lds_dma16_s<j * 1024>(b, voff, lds_wave + SLOT * kChunkBytes);
            });
        } else {
            asm volatile("" : "+v"(base));
            // one 64-bit address per chunk; the 1 KiB steps ride in the instruction's immediate offset, which the hardware
            // adds to the global address AND to the LDS destination
            const char* b = base + OFF;
            static_for<C::GPC>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                __builtin_amdgcn_global_load_lds(TGTC_GPTR(b), TGTC_LPTR(lds_wave + SLOT * kChunkBytes), 16, j * 1024, 0);
            });
        }
    }
    template <int CH>
    __device__ __forceinline__ void issue() const {
        if constexpr (CH < PADC) {
            constexpr int seg = seg_of(CH);
            issue_chunk<(size_t)(CH - Map::chunk0(seg)) * kChunkBytes, CH % C::SLOTS>(src[seg]);
        } else if constexpr (PERSIST) {
            static_assert(CH < PADC + C::SLOTS, "look-ahead beyond the next pass's first ring");
            issue_chunk<(size_t)(CH - PADC) * kChunkBytes, CH % C::SLOTS>(next);
        }
    }
    // PERSIST: the per-lane source pointer of a packed stream for this wave / lane
    __device__ __forceinline__ static const char* lane_src(const char* stream, int wave, int lane) {
        if constexpr (ASM_DMA) return stream;
        else return stream + wave * (C::GPC * 1024) + lane * 16;
    }
    // PERSIST, once per kernel: chunks 0 .. SLOTS-2 of the first stream (`next`), the state every enter() expects
    __device__ __forceinline__ void persist_prologue() const {
        static_for<LOOK>([&](auto ch) { issue<PADC + decltype(ch)::value>(); });
    }
    // PERSIST: enter the stream `next` points to (virtual chunk PADC of the pass that ends = chunk 0 of the new
    // one); the caller sets `next` again before the pass issues its first look-ahead into the following stream.
    __device__ __forceinline__ void enter_ring() {
        static_assert(PERSIST, "enter_ring() belongs to persistent streams");
        wait_vmcnt<(LOOK - 2) * C::GPC>();
        __builtin_amdgcn_s_barrier();
        if constexpr (Map::NSEG == 1) src[0] = next;
        issue<LOOK>();
    }
    __device__ __forceinline__ void enter() {
        enter_ring();
        static_for<G>([&](auto f) { fetch<decltype(f)::value>(); });
        __builtin_amdgcn_sched_barrier(0);
    }
    // PERSIST: walk the dummy chunks behind the last consumed one (CH0 = first chunk index not entered yet)
    template <int CH0>
    __device__ __forceinline__ void finish() const {
        static_assert(PERSIST, "finish() belongs to persistent streams");
        static_for<(PADC > CH0 ? PADC - CH0 : 0)>([&](auto i) { boundary<CH0 + decltype(i)::value>(); });
    }
    __device__ __forceinline__ void prologue() const {
        static_for<C::SLOTS>([&](auto ch) { issue<decltype(ch)::value>(); });
    }
    template <int FRAG, int PART>
    __device__ __forceinline__ half8 read() const {
        constexpr int off = ((FRAG / C::FPC) % C::SLOTS) * kChunkBytes + (FRAG % C::FPC) * C::FRAG_BYTES + PART * 1024;
        typedef __attribute__((address_space(3))) const half8* lds_h8;
        if constexpr (off < 65536) return *(lds_h8)(lane_lo + off);
        else return *(lds_h8)(lane_hi + (off - 65536));
    }
    template <int F>
    __device__ __forceinline__ void fetch() {
        if constexpr (F < NFRAG && (!(kAbl & 4) || F < 2 * G)) {
            qh[(F / G) & 1][F % G] = read<F, 0>();
            if constexpr (C::SPLIT) ql[(F / G) & 1][F % G] = read<F, 1>();
        }
    }
    // wait for this wave's pieces of chunks 0 and 1, then the workgroup barrier that makes them visible
    __device__ __forceinline__ void start_ring() const {
        constexpr int issued_last = (C::SLOTS - 1 < NCHUNK - 1) ? C::SLOTS - 1 : NCHUNK - 1;
        constexpr int need = NCHUNK > 1 ? 1 : 0;
        wait_vmcnt<(issued_last - need) * C::GPC>();
        __builtin_amdgcn_s_barrier();
    }
    __device__ __forceinline__ void start() {
        start_ring();
        static_for<G>([&](auto f) { fetch<decltype(f)::value>(); });
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int CH>
    __device__ __forceinline__ void boundary() const {
        if constexpr (PERSIST) {
            wait_vmcnt<(LOOK - 2) * C::GPC>();
            __builtin_amdgcn_s_barrier();
            issue<CH + LOOK>();
        } else if constexpr (CH + 1 < NCHUNK) {
            constexpr int issued_last = (CH + C::SLOTS - 2 < NCHUNK - 1) ? CH + C::SLOTS - 2 : NCHUNK - 1;
            if constexpr (!(kAbl & 1)) wait_vmcnt<(issued_last - (CH + 1)) * C::GPC>();
            if constexpr (!(kAbl & 2)) __builtin_amdgcn_s_barrier();
            if constexpr (!(kAbl & 1)) issue<CH + C::SLOTS - 1>();
        }
    }
    // fragment F (hi [+lo]).  On the first fragment of a group the group's reads are retired (hipcc only emits
    // lgkmcnt(0), so every outstanding read must be old by then); the NEXT group's reads are issued two
    // fragments per window over the first half of the current group, so that each ds_read hides in an MFMA's
    // issue shadow and the youngest one is >= G/2 windows old at the next wait.  The caller closes the window
    // with close_window<F>().
    template <int F>
    __device__ __forceinline__ void get(half8& ah, half8& al) {
        constexpr int buf = (F / G) & 1, k = F % G;
        if constexpr (k == 0) {
            if constexpr (STAG == 0) {
                if constexpr (F % C::FPC == 0 && F > 0) boundary<F / C::FPC>();
            } else {
                // waves 0..3 enter chunk c at its first fragment; waves 4..7 run the same boundary STAG groups earlier
                // (at fragment c*FPC - STAG*G, or with the first group where that is negative)
                static_assert(STAG * G <= C::FPC, "a late wave must not fall behind the chunk that is re-filled");
                constexpr bool early_here = F % C::FPC == 0 && F > 0;
                constexpr int FL = F + STAG * G;
                constexpr bool late_here = F > 0 ? (FL % C::FPC == 0 && FL / C::FPC < NCHUNK) : (STAG * G == C::FPC && NCHUNK > 1);
                if constexpr (early_here || late_here) {
                    if (late) {
                        if constexpr (late_here) boundary<(F > 0 ? FL / C::FPC : 1)>();
                    } else {
                        if constexpr (early_here) boundary<F / C::FPC>();
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {  // a "use": the compiler's lgkmcnt wait lands HERE
                asm volatile("" ::"v"(qh[buf][j]));
                if constexpr (C::SPLIT) asm volatile("" ::"v"(ql[buf][j]));
            }
        }
        ah = qh[buf][k];
        if constexpr (C::SPLIT) al = ql[buf][k];
        if constexpr (2 * k < G) {
            fetch<F - k + G + 2 * k>();
            fetch<F - k + G + 2 * k + 1>();
        }
    }
    // Pin the window's instruction mix: after every MFMA one pending ds_read (if any) and up to two VALU
    // instructions -- what fits into the ~8 spare issue cycles of a 16x16x32 MFMA -- then fence the window.
    template <int F, int N_MFMA, int EXTRA_READS, int N_VALU = 2>
    __device__ __forceinline__ void close_window() const {
        constexpr int NR = reads_in_window(F) + EXTRA_READS;
        static_for<N_MFMA>([&](auto i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            if constexpr (decltype(i_)::value < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, N_VALU, 0);  // VALU
        });
        __builtin_amdgcn_sched_barrier(0);
    }
    // consume fragments [F0, F0+N) without using them (alignment gaps between segments)
    template <int F0, int N>
    __device__ __forceinline__ void skip() {
        static_for<N>([&](auto i) {
            half8 a, b;
            get<F0 + decltype(i)::value>(a, b);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
};

__device__ __forceinline__ float4v mfma16(half8 a, half8 b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// One dense layer.  B operands: Bh[KS][NCT] (+ Bl in split mode).  Fragment (rt,ks) is stream
// fragment FRAG0 + rt*KS + ks.  bias_lane = LDS address of (bias table) + 4*(lane>>4) floats.
// epi(ic<rt>, ic<c>, ic<half>, acc) receives one half (registers 2*half, 2*half+1) of the fp32 tile of row
// tile rt, column tile c:  rows 16*rt + 4*(lane>>4) + r, column = sample lane&15.
// The epilogue of row tile rt is issued in 2*NCT small units spread evenly behind the MFMAs of row
// tile rt+1 (two accumulator sets): a 16x16x32 MFMA leaves ~8 of its 16 cycles of issue bandwidth, i.e.
// one or two VALU instructions, so the ReLU / convert work must trickle, not burst.
template <class C, int FRAG0, int KS, int RT, int BIAS0, class StreamT, class Epi>
__device__ __forceinline__ void dense_layer(StreamT& st, lds_cptr bias_lane, const half8 (&Bh)[KS][C::NCT],
                                            const half8 (&Bl)[KS][C::NCT], Epi&& epi) {
    constexpr int NCT = C::NCT;
    constexpr int UNITS = 2 * NCT;
    constexpr int PER = (UNITS + KS - 1) / KS;  // epilogue units per k-step window
    typedef __attribute__((address_space(3))) const float4v* lds_f4;
    float4v acc[2][NCT];
    float4v bias[2];  // read one row tile ahead so that its wait never lands on a fresh burst
    bias[0] = *(lds_f4)(bias_lane + BIAS0 * 4);
    static_for<RT>([&](auto rt_) {
        constexpr int rt = decltype(rt_)::value;
        constexpr int cur = rt & 1;
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[cur][c] = bias[cur];
        static_for<KS>([&](auto ks_) {
            constexpr int ks = decltype(ks_)::value;
            half8 ah, al;
            st.template get<FRAG0 + rt * KS + ks>(ah, al);
            if constexpr (ks == 0 && rt + 1 < RT) bias[cur ^ 1] = *(lds_f4)(bias_lane + (BIAS0 + 16 * (rt + 1)) * 4);
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[cur][c] = mfma16(ah, Bh[ks][c], acc[cur][c]);
            if constexpr (C::SPLIT) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[cur][c] = mfma16(al, Bh[ks][c], acc[cur][c]);
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[cur][c] = mfma16(ah, Bl[ks][c], acc[cur][c]);
            }
            if constexpr (rt > 0) {
                static_for<PER>([&](auto p_) {
                    constexpr int u = ks * PER + decltype(p_)::value;
                    if constexpr (u < UNITS) epi(ic<rt - 1>{}, ic<u / 2>{}, ic<u % 2>{}, acc[cur ^ 1][u / 2]);
                });
            }
            st.template close_window<FRAG0 + rt * KS + ks, NCT * (C::SPLIT ? 3 : 1), (ks == 0 && rt + 1 < RT) ? 1 : 0>();
        });
    });
    static_for<UNITS>([&](auto u_) {
        constexpr int u = decltype(u_)::value;
        epi(ic<RT - 1>{}, ic<u / 2>{}, ic<u % 2>{}, acc[(RT - 1) & 1][u / 2]);
    });
}

__device__ __forceinline__ float relu(float v) {
    // integer max: one v_max_i32, no NaN-canonicalising v_max_f32 pair
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0));
}

// ReLU and hi/lo split of a pair of fp32 values into two packed fp16 pairs: hi = fp16(relu(v)) (one v_cvt_pk),
// lo = fp16(relu(v) - hi) straight out of v_fma_mixlo/mixhi_f16 (v * 1.0 - h in fp32, rounded once; v - h is exact in
// fp32, so this is bit-identical to (half)(v - (float)h)): five VALU instructions per pair instead of eight.
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hpk, unsigned& lpk) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    v0 = relu(v0), v1 = relu(v1);
    hpk = __builtin_bit_cast(unsigned, __builtin_convertvector((float2v{v0, v1}), half2v));
    // (mixlo leaves the upper half of its destination alone, mixhi then writes it: no zero-initialisation needed)
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lpk) : "v"(v0), "v"(hpk));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lpk) : "v"(v1), "v"(hpk));
}
__device__ __forceinline__ void set_pair(half8& v, int e0, unsigned pk) {
    u4 r = __builtin_bit_cast(u4, v);
    r[e0 >> 1] = pk;
    v = __builtin_bit_cast(half8, r);
}

// ReLU + fp16 (hi/lo) conversion of one HALF (registers 2*HALF, 2*HALF+1) of an accumulator tile into the
// next layer's B fragment: output row tile rt feeds k-step rt/2, elements (rt&1)*4 + r.
// fp16 mode: convert the pair first (v_cvt_pk_f16_f32), then one packed max (v_pk_max_f16) -- rounding
// is monotonic and sign preserving, so relu(cvt(x)) == cvt(relu(x)).
template <class C, int RT_IDX, int HALF>
__device__ __forceinline__ void store_act(const float4v& acc, half8& yh, half8& yl) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    constexpr int e0 = (RT_IDX & 1) * 4 + 2 * HALF;
    if constexpr (kAbl & 8) {
        yh[e0] = (half_t)acc[2 * HALF];
        if constexpr (C::SPLIT) yl[e0] = yh[e0];
    } else if constexpr (!C::SPLIT) {
        half2v h = __builtin_convertvector((float2v{acc[2 * HALF], acc[2 * HALF + 1]}), half2v);
        h = __builtin_elementwise_max(h, (half2v{(half_t)0, (half_t)0}));
        yh[e0] = h[0], yh[e0 + 1] = h[1];
    } else {
        unsigned hpk, lpk;
        split_pair(acc[2 * HALF], acc[2 * HALF + 1], hpk, lpk);
        set_pair(yh, e0, hpk);
        set_pair(yl, e0, lpk);
    }
}

// ReLU + fp16 (hi/lo) conversion of one HALF of an accumulator tile into packed 32-bit registers (two halves each),
// the same values store_act writes into elements e0, e0+1 of the next layer's B fragment.
template <bool SPLIT, int HALF>
__device__ __forceinline__ void pack_act(const float4v& acc, unsigned& ph, unsigned& pl) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    if constexpr (!SPLIT) {
        half2v h = __builtin_convertvector((float2v{acc[2 * HALF], acc[2 * HALF + 1]}), half2v);
        h = __builtin_elementwise_max(h, (half2v{(half_t)0, (half_t)0}));
        ph = __builtin_bit_cast(unsigned, h);
    } else {
        half2v h, l;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float v = relu(acc[2 * HALF + r]);
            h[r] = (half_t)v;
            l[r] = (half_t)(v - (float)h[r]);
        }
        ph = __builtin_bit_cast(unsigned, h), pl = __builtin_bit_cast(unsigned, l);
    }
}

// Move a 32-bit value into / out of the accumulator half of the register file.  The empty asm statements only
// pin the register class at that point; hipcc emits the v_accvgpr_write / v_accvgpr_read copies itself (and
// their hazard wait states).
__device__ __forceinline__ unsigned park(unsigned v) {
    asm("" : "+a"(v));
    return v;
}
__device__ __forceinline__ unsigned unpark(unsigned a) {
    asm("" : "+v"(a));
    return a;
}
__device__ __forceinline__ half8 unpark4(const unsigned (&p)[4]) {
    u4 r{unpark(p[0]), unpark(p[1]), unpark(p[2]), unpark(p[3])};
    return __builtin_bit_cast(half8, r);
}
__device__ __forceinline__ void park4(half8 v, unsigned (&p)[4]) {
    const u4 r = __builtin_bit_cast(u4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = park(r[i]);
}

// ------------------------------------------------------------------------------------------------
// Positional encoding straight into B fragments.
//
// k-slot assignment (mirrored by the packer, see pe63_col / pe27_col): lane group g = lane>>4 owns,
// per sample, 8 (sin,cos) pairs q = 8g+i (i = 0..7) of the 30 (band,coord) pairs of the 63-wide
// encoding; pair i lives in k-step i>>2, elements 2*(i&3) (sin) and 2*(i&3)+1 (cos).  Pairs 30 and 31
// carry the raw coordinates (x,y) and (z,0).
constexpr double kInv2Pi = 0.15915494309189533576888376337251436;

template <bool ACCURATE>
__device__ __forceinline__ void sincos_turns(double turns, float& s, float& c) {
    // sin/cos(2*pi*turns) with the range reduction done in float64 (the reference encodes in float64,
    // models.py:46-60 on float64 points; an fp32 argument of up to ~770 rad would lose 4-5 digits).
    if constexpr (ACCURATE) {
        const double t = turns * 4.0;
        const double n = rint(t);
        // (the constant is materialised here: as a plain literal hipcc pairs it into a v_pk_mul_f32 operand that it keeps in
        // a register for the whole kernel -- and spills and reloads around the asm blocks of the fp16mx pass)
        float half_pi = 1.57079632679489661923f;
        asm volatile("" : "+v"(half_pi));
        const float a = (float)(t - n) * half_pi;  // [-pi/4, pi/4]
        const float a2 = a * a;
        const float sp = a + a * a2 * (-1.6666654611e-1f + a2 * (8.3321608736e-3f + a2 * (-1.9515295891e-4f)));
        const float cp = 1.0f - 0.5f * a2 +
                         a2 * a2 * (4.166664568298827e-2f + a2 * (-1.388731625493765e-3f + a2 * 2.443315711809948e-5f));
        const int qd = (int)n & 3;
        float ss = (qd & 1) ? cp : sp, cc = (qd & 1) ? sp : cp;
        if (qd == 1) cc = -cc;
        if (qd == 2) ss = -ss, cc = -cc;
        if (qd == 3) ss = -ss;
        s = ss, c = cc;
    } else {
        const float f = (float)(turns - rint(turns));  // [-0.5, 0.5] turns; v_sin/v_cos take turns
        s = __builtin_amdgcn_sinf(f);
        c = __builtin_amdgcn_cosf(f);
    }
}

// elements j0, j0+1 (one 32-bit register) of an encoding's hi / lo B fragments.  SPLIT: the pair's hi halves by one packed
// convert, the lo halves straight from v_fma_mixlo/mixhi_f16 (v * 1.0 - h in fp32, rounded once: the same bits as
// (half)(v - (float)h)) -- three instructions per pair instead of the eight the scalar form compiles to; the encoders run
// once per pass in front of the generated streams, where a lone wave has nobody to hide them (profiles/r4_kernel_variants.md).
template <bool SPLIT>
__device__ __forceinline__ void put_pair(half8& hi, half8& lo, int j0, float s, float c) {
    if constexpr (SPLIT) {
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        typedef float float2v __attribute__((ext_vector_type(2)));
        unsigned hpk = __builtin_bit_cast(unsigned, __builtin_convertvector((float2v{s, c}), half2v)), lpk;
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lpk) : "v"(s), "v"(hpk));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lpk) : "v"(c), "v"(hpk));
        set_pair(hi, j0, hpk);
        set_pair(lo, j0, lpk);
    } else {
        hi[j0] = (half_t)s, hi[j0 + 1] = (half_t)c;
    }
}

// logical column (0..62) of pair q, function fn (0 sin, 1 cos) in the reference encoding order
// [x, sin(2^0 x), cos(2^0 x), ...] (models.py:50-57); -1 = padding.
__host__ __device__ inline int pe_col(int q, int fn, int n_pairs) {
    if (q < n_pairs) return 3 + 6 * (q / 3) + 3 * fn + q % 3;
    if (q == n_pairs) return fn;            // (x, y)
    if (q == n_pairs + 1) return fn == 0 ? 2 : -1;  // (z, pad)
    return -1;
}

// 63-wide point encoding -> 2 k-steps.  enc_out (optional) receives the float32 encoding [63].
template <bool SPLIT, bool ACCURATE>
__device__ __forceinline__ void encode_point(const double (&p)[3], int g, half8 (&hi)[2], half8 (&lo)[2],
                                             float* enc_out) {
    const double rx = p[0] * kInv2Pi, ry = p[1] * kInv2Pi, rz = p[2] * kInv2Pi;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = 8 * g + i;
        const int band = q / 3, a = q - 3 * band;
        const double r = a == 0 ? rx : (a == 1 ? ry : rz);
        float s, c;
        sincos_turns<ACCURATE>(ldexp(r, band), s, c);
        if (i >= 6) {  // only lane group 3 reaches the raw pairs 30, 31
            if (q == 30) s = (float)p[0], c = (float)p[1];
            if (q == 31) s = (float)p[2], c = 0.0f;
        }
        put_pair<SPLIT>(hi[i >> 2], lo[i >> 2], 2 * (i & 3), s, c);
        if (enc_out) {
            const int cs = pe_col(q, 0, 30), cc = pe_col(q, 1, 30);
            enc_out[cs] = s;
            if (cc >= 0) enc_out[cc] = c;
        }
    }
}

// 27-wide direction encoding -> 1 k-step: lane group g owns pairs q = 4g+i of the 12 (band,coord)
// pairs; pairs 12, 13 carry (x,y), (z,0); 14, 15 are padding.
template <bool SPLIT, bool ACCURATE>
__device__ __forceinline__ void encode_dir(const double (&d)[3], int g, half8& hi, half8& lo, float* enc_out) {
    const double rx = d[0] * kInv2Pi, ry = d[1] * kInv2Pi, rz = d[2] * kInv2Pi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 4 * g + i;
        const int band = q / 3, a = q - 3 * band;
        const double r = a == 0 ? rx : (a == 1 ? ry : rz);
        float s, c;
        sincos_turns<ACCURATE>(ldexp(r, band), s, c);
        if (q == 12) s = (float)d[0], c = (float)d[1];
        if (q == 13) s = (float)d[2], c = 0.0f;
        if (q >= 14) s = 0.0f, c = 0.0f;
        put_pair<SPLIT>(hi, lo, 2 * i, s, c);
        if (enc_out && q < 14) {
            const int cs = pe_col(q, 0, 12), cc = pe_col(q, 1, 12);
            enc_out[cs] = s;
            if (cc >= 0) enc_out[cc] = c;
        }
    }
}

// The same fragments from an already-encoded float32 row (MLP_style.forward boundary).
template <bool SPLIT>
__device__ __forceinline__ void load_encoded_point(const float* enc, int g, half8 (&hi)[2], half8 (&lo)[2]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = 8 * g + i;
        const int cs = pe_col(q, 0, 30), cc = pe_col(q, 1, 30);
        put_pair<SPLIT>(hi[i >> 2], lo[i >> 2], 2 * (i & 3), enc[cs], cc >= 0 ? enc[cc] : 0.0f);
    }
}
template <bool SPLIT>
__device__ __forceinline__ void load_encoded_dir(const float* enc, int g, half8& hi, half8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 4 * g + i;
        const int cs = pe_col(q, 0, 12), cc = pe_col(q, 1, 12);
        put_pair<SPLIT>(hi, lo, 2 * i, cs >= 0 ? enc[cs] : 0.0f, cc >= 0 ? enc[cc] : 0.0f);
    }
}

// a 32-wide float vector (the style latent) -> 1 k-step in natural order k = 8g + j
template <bool SPLIT>
__device__ __forceinline__ void load_vec32(const float* v, int g, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[8 * g + j];
        const half_t h = (half_t)x;
        hi[j] = h;
        if constexpr (SPLIT) lo[j] = (half_t)(x - (float)h);
    }
}

// ------------------------------------------------------------------------------------------------
// k-slot -> logical input column maps used by the packer (host).  `g` = lane>>4, `j` = element.
// activations produced by a previous layer's accumulator tiles:
__host__ __device__ inline int act_col(int ks, int g, int j) { return 32 * ks + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)); }
__host__ __device__ inline int pe63_col(int ks, int g, int j) { return pe_col(8 * g + 4 * ks + (j >> 1), j & 1, 30); }
__host__ __device__ inline int pe27_col(int g, int j) { return pe_col(4 * g + (j >> 1), j & 1, 12); }
__host__ __device__ inline int vec32_col(int g, int j) { return 8 * g + j; }

}  // namespace tgtc
