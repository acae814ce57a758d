// Shared pieces of the fused ray kernels (render_fused.hip: the plain render; render_styled_fused.hip: the stylised one):
// launch arguments, per-precision geometry, the persistent stream types, one NeRF pass over a wave's tiles, the per-wave
// LDS strip.  The design is described at the top of render_fused.hip.
#pragma once
#include "common.h"
#include "render_args.h"
#include "mlp_core.h"
#ifndef TGTC_FUSED_X3_ASM_DMA
#define TGTC_FUSED_X3_ASM_DMA false   // fp16 / fp16x3 passes: builtin LDS-DMA (asm + SGPR-base addressing measured, profiles section 16)
#endif
#include "mlp_layouts.h"
#include "mlp_mx.h"
#include "mlp_nerf_chain.h"
#include "mlp_nerf_mx_chain.h"
#include "mlp_pack.h"
#include "raymarch_wave.h"

namespace tgtc {


constexpr int kFusedDepthsOnly = -1;                      // PF of the kernel that stops after the fine sampling (stylised render)
constexpr int kFusedMaxTotal = 256;                       // Nc + Nf supported by the per-wave LDS strip
constexpr int kFusedStripBytes = (kFusedMaxTotal + 192 + 8) * 4;   // depths | weights / cdf (Nc <= 192) | compositing state

// Geometry per precision: 8 waves = 2 per SIMD (mlp_nerf.hip CfgFast / CfgExact, mlp_nerf_mx.hip CfgMx)
// (SLOTS: 16 KiB ring slots; the stylised kernel gives one up for its second bias table)
template <int PREC, int SLOTS = kRingSlots>
struct FusedCfg;
template <int SLOTS>
struct FusedCfg<TGTC_PREC_FP16X3, SLOTS> {
    using C = MlpCfg<8, 1, true, 4, SLOTS>;
};
template <int SLOTS>
struct FusedCfg<TGTC_PREC_FP16, SLOTS> {
    using C = MlpCfg<8, 2, false, 4, SLOTS>;
};
template <int SLOTS>
struct FusedCfg<TGTC_PREC_FP16_FP6, SLOTS> {
    using C = MlpCfg<8, 1, false, 4, SLOTS>;
};

template <int PREC, bool FULL, int SLOTS = kRingSlots>
struct FusedStream {
    using C = typename FusedCfg<PREC, SLOTS>::C;
    using type = WeightStream<C, SingleStreamMap<(FULL ? NerfLayout::kFragsFull : NerfLayout::kFragsSigma)>, true, TGTC_FUSED_X3_ASM_DMA>;
};
template <bool FULL, int SLOTS>
struct FusedStream<TGTC_PREC_FP16_FP6, FULL, SLOTS> {
    using C = typename FusedCfg<TGTC_PREC_FP16_FP6, SLOTS>::C;
    using type = MxReader<C, SingleStreamMap<nerf_mx_units(FULL)>, kNerfMxTable, true>;
};

__device__ __forceinline__ double uniform_f64(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// One pass: the network of one precision over this wave's NCT tiles (depths tt[c]) of its ray (o, d).
// sig[c] / col[c][0..2] are valid in lanes 0..15 (sample = lane & 15).  Everything per-lane the pass needs (stream
// reader, LDS bases) is rebuilt here from the wave / lane ids, so that nothing but scalars lives across the ~250
// registers of the layer chain (values that do are spilled to scratch, and a scratch reload waits vmcnt(0), i.e. for
// the whole look-ahead of the ring).
template <int PREC, bool FULL, int SLOTS = kRingSlots>
__device__ __forceinline__ void fused_pass(char* smem, int wave, int lane, const char* cur_net, const char* next_stream,
                                           const double (&o)[3], const double (&d)[3],
                                           const float (&tt)[FusedCfg<PREC>::C::NCT], float (&sig)[FusedCfg<PREC>::C::NCT],
                                           float (&col)[FusedCfg<PREC>::C::NCT][3]) {
    using C = typename FusedCfg<PREC, SLOTS>::C;
    using StreamT = typename FusedStream<PREC, FULL, SLOTS>::type;
    constexpr int NCT = C::NCT;
    constexpr bool SPLIT = PREC != TGTC_PREC_FP16;   // hi + lo encodings (fp16x3 and the fp16+fp6 PE k-steps)
    const int g = lane >> 4, n = lane & 15;
    half8 pe_h[2][NCT], pe_l[2][NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        double p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = o[k] + (double)tt[c] * d[k];   // rendering.py:27 / utils.py:529,578
        half8 h2[2], l2[2];
        encode_point<SPLIT, SPLIT>(p, g, h2, l2, nullptr);
        pe_h[0][c] = h2[0], pe_h[1][c] = h2[1], pe_l[0][c] = l2[0], pe_l[1][c] = l2[1];
    }
    auto colour = [&](auto c_, auto h_, const float4v& acc) {
        constexpr int c = decltype(c_)::value, hf = decltype(h_)::value;
#pragma unroll
        for (int r = 2 * hf; r < (hf ? 3 : 2); ++r) col[c][r] = 1.0f / (1.0f + expf(-acc[r]));   // models.py:111
    };
    const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * g);
    StreamT st;
    const char* const one[1] = {cur_net + kNerfBiasBytes};
    st.init(one, smem, wave, lane);
    const char* const next_src = StreamT::lane_src(next_stream, wave, lane);   // the packed stream the look-ahead runs into
    if constexpr (PREC == TGTC_PREC_FP16_FP6) {
        constexpr int NQ = nerf_mx_groups(FULL);
        const lds_cptr rs_lane = opaque((lds_cptr)smem + C::RING_BYTES + kNerfMxScaleOff + 2 * n);
        if constexpr (TGTC_MX_ASM && FULL) st.relane(smem, wave);   // (per-lane addresses from a lane id read HERE: see MxReader::relane)
        st.ring.next = st.ring.src[0];   // the stream this pass enters (its first chunks are in flight) ...
        st.template enter<0, NQ>();
        st.ring.next = next_src;         // ... and the one its look-ahead runs into
        const half8 Ph[2] = {pe_h[0][0], pe_h[1][0]}, Pl[2] = {pe_l[0][0], pe_l[1][0]};
#if TGTC_MX_ASM == 2
        if constexpr (FULL && C::SLOTS == 8 && kChunkBytes == 16384) {
            // the whole pass as ONE generated instruction stream (tools/gen_mx_asm.py, mx_asm_nerf.inc): the encodings go in
            // pinned, sigma and the colour head's accumulator come out; nothing else of this function lives across it
            half8 keep[6] = {Ph[0], Ph[1], Pl[0], Pl[1], half8{}, half8{}};
            encode_dir<true, true>(d, g, keep[4], keep[5], nullptr);
            st.relane(smem, wave);
            const int fl = fresh_lane_id();
            const lds_cptr bl = opaque((lds_cptr)smem + C::RING_BYTES + 16 * (fl >> 4));
            const lds_cptr rl = opaque((lds_cptr)smem + C::RING_BYTES + kNerfMxScaleOff + 2 * (fl & 15));
            float sigma, rgb[3];
            mx_asm_nerf_full_pass(st, bl, rl, keep, sigma, rgb);
            st.relane(smem, wave);
            sig[0] = sigma;
#pragma unroll
            for (int r = 0; r < 3; ++r) col[0][r] = 1.0f / (1.0f + expf(-rgb[r]));   // models.py:111
        } else
#endif
        nerf_chain_mx<C, FULL>(
            st, bias_lane, rs_lane, Ph, Pl, [&](half8& dh, half8& dl) { encode_dir<true, true>(d, g, dh, dl, nullptr); },
            [&](float s) { sig[0] = s; }, [](auto, auto, const float4v&) {},
            [&](auto h_, const float4v& acc) { colour(ic<0>{}, h_, acc); }, smem, wave);
        st.template finish<NQ>();
    } else {
        st.next = st.src[0];
        st.enter();
        st.next = next_src;
        nerf_chain<C, FULL>(
            st, bias_lane, pe_h, pe_l, [&](auto, half8& dh, half8& dl) { encode_dir<SPLIT, SPLIT>(d, g, dh, dl, nullptr); },
            [&](auto c_, float s) { sig[decltype(c_)::value] = s; }, [](auto, auto, auto, const float4v&) {}, colour);
        st.template finish<StreamT::NCHUNK>();
    }
}

// per-wave LDS strip: depths | weights / cdf | compositing state
constexpr int kStripAll = 0, kStripW = kFusedMaxTotal, kStripAcc = kFusedMaxTotal + 192;

}  // namespace tgtc
