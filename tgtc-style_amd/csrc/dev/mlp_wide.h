// The "wide" formulation of the fused MLP kernels: v_mfma_f32_32x32x16_f16, one wave per SIMD.
//
// Why (profiles/r2_kernel_variants.md section 4): a 16x16x32 MFMA occupies the matrix pipe for 16 cycles and the SIMD's
// vector issue for 8 of them; what is left are two VALU-class slots per MFMA for BOTH waves of the SIMD, and the
// fp16x3 / fp16mx layer loops need more than that (LDS fragment reads, ReLU / hi-lo split, LDS-DMA issue, waits).
// A 32x32x16 MFMA does twice the work per instruction: 32 cycles of pipe for the same 8 cycles of issue, i.e. six
// free slots per MFMA, enough for ONE wave per SIMD to keep the pipe fed by itself.  That wave gets the whole
// 512-register file, so both 256-feature activation sets of 32 samples (hi + lo: 128 registers each) fit.
//
// Layout.  Layers stay transposed (mlp_core.h): D[32 features x 32 samples] += W[32 x 16] * H^T[16 x 32].
//   A operand: lane (r = lane&31, h = lane>>5) holds W[32*T + r][k-slot 8h + j], j = 0..7   (1 KiB fragment, from the ring)
//   B operand: lane (r, h) holds input k-slot 8h + j of sample r
//   C/D:       lane (r, h), register q (0..15) = feature 32*T + (q&3) + 8*(q>>2) + 4*h of sample r
// so registers 8s .. 8s+7 of output tile T are, after ReLU and conversion, the B fragment of k-step 2T + s of the next
// layer, for the k-permutation act_col_w() that the packer folds into the weight columns.  Nothing moves between lanes.
//
// The weight stream has the same granules as the 16x16 one (1 KiB fragments, hi [+ lo], 16 KiB chunks through the
// 128 KiB ring); only the order of rows / columns inside a fragment differs, so WeightStream is shared.
#pragma once
#include "mlp_core.h"

namespace tgtc {

typedef float float16v __attribute__((ext_vector_type(16)));

// 4 waves (one per SIMD), 32 samples per wave
template <bool SPLIT, int G = 4, int SLOTS = kRingSlots>
using WideCfg = MlpCfg<4, 2, SPLIT, G, SLOTS, 1, false>;

__device__ __forceinline__ float16v mfma32(half8 a, half8 b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ---- k-slot -> logical input column (packer and kernels agree through these)
__host__ __device__ inline int act_col_w(int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
__host__ __device__ inline int pe63_col_w(int ks, int h, int j) { return pe_col(16 * h + 4 * ks + (j >> 1), j & 1, 30); }
__host__ __device__ inline int pe27_col_w(int ks, int h, int j) { return pe_col(8 * h + 4 * ks + (j >> 1), j & 1, 12); }

// ---- compile-time layout of the packed NeRF network (wide): 16-deep k-steps, 32-row tiles
//   layer      L0  L1  L2  L3  L4  L5  L6  L7  SIG REMAP C0  C1
constexpr int kNerfKSW[12] = {4, 16, 16, 16, 16, 20, 16, 16, 16, 16, 18, 8};
constexpr int kNerfRTW[12] = {8, 8, 8, 8, 8, 8, 8, 8, 1, 8, 4, 1};
constexpr int nerf_frag0_w(int l) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += kNerfKSW[i] * kNerfRTW[i];
    return f;
}
constexpr int nerf_bias0_w(int l) {
    int b = 0;
    for (int i = 0; i < l; ++i) b += 32 * kNerfRTW[i];
    return b;
}
struct NerfLayoutW {
    static constexpr int frag0(int l) { return nerf_frag0_w(l); }
    static constexpr int bias0(int l) { return nerf_bias0_w(l); }
    static constexpr int kFragsSigma = nerf_frag0_w(9);
    static constexpr int kFragsFull = nerf_frag0_w(12);
    static constexpr int kBiasFloats = nerf_bias0_w(12);
};
static_assert(NerfLayoutW::kFragsFull == 1184 && NerfLayoutW::kFragsSigma == 976, "wide fragment counts");
static_assert(NerfLayoutW::kBiasFloats * 4 <= 16384, "wide bias table");

// One dense layer.  Bh[KS] (+ Bl): the B fragments of this wave's 32 samples.  Fragment (rt, ks) is stream fragment
// FRAG0 + rt*KS + ks.  bias_lane = LDS address of the bias table + 64*(lane>>5) bytes; a row tile's 32 biases are stored
// [h][q] so that four 16-byte reads fill the accumulator in register order.
// epi(ic<rt>, ic<u>, acc): pair u (registers 2u, 2u+1) of the finished tile rt; the eight pairs of tile rt are handed
// out evenly behind the MFMAs of tile rt+1 (two accumulator sets), so the epilogue trickles through the issue slots.
template <class C, int FRAG0, int KS, int RT, int BIAS0, class StreamT, class Epi>
__device__ __forceinline__ void dense_layer_w(StreamT& st, lds_cptr bias_lane, const half8 (&Bh)[KS], const half8 (&Bl)[KS],
                                              Epi&& epi) {
    constexpr int UNITS = 8;
    typedef __attribute__((address_space(3))) const float4v* lds_f4;
    float16v acc[2];
    float4v bias[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias[0][q] = *(lds_f4)(bias_lane + BIAS0 * 4 + q * 16);
    static_for<RT>([&](auto rt_) {
        constexpr int rt = decltype(rt_)::value;
        constexpr int cur = rt & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[cur][4 * q + i] = bias[cur][q][i];
        static_for<KS>([&](auto ks_) {
            constexpr int ks = decltype(ks_)::value;
            half8 ah, al;
            st.template get<FRAG0 + rt * KS + ks>(ah, al);
            if constexpr (ks == 0 && rt + 1 < RT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bias[cur ^ 1][q] = *(lds_f4)(bias_lane + (BIAS0 + 32 * (rt + 1)) * 4 + q * 16);
            }
            acc[cur] = mfma32(ah, Bh[ks], acc[cur]);
            if constexpr (C::SPLIT) {
                acc[cur] = mfma32(al, Bh[ks], acc[cur]);
                acc[cur] = mfma32(ah, Bl[ks], acc[cur]);
            }
            if constexpr (rt > 0) {
                static_for<(ks + 1) * UNITS / KS - ks * UNITS / KS>([&](auto p_) {
                    constexpr int u = ks * UNITS / KS + decltype(p_)::value;
                    epi(ic<rt - 1>{}, ic<u>{}, acc[cur ^ 1]);
                });
            }
            st.template close_window<FRAG0 + rt * KS + ks, C::SPLIT ? 3 : 1, (ks == 0 && rt + 1 < RT) ? 4 : 0, 4>();
        });
    });
    static_for<UNITS>([&](auto u_) { epi(ic<RT - 1>{}, u_, acc[(RT - 1) & 1]); });
}

// ReLU + fp16 (hi/lo) conversion of pair U of output tile RT_IDX into the next layer's B fragments Y[2*RT_IDX + (U>>2)]
template <class C, int U>
__device__ __forceinline__ void store_act_w(const float16v& acc, half8& yh, half8& yl) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    constexpr int e0 = 2 * (U & 3);
    if constexpr (kAbl & 8) {   // timing ablation: one instruction per pair
        yh[e0] = (half_t)acc[2 * U];
        if constexpr (C::SPLIT) yl[e0] = yh[e0];
    } else if constexpr (!C::SPLIT) {
        half2v h = __builtin_convertvector((float2v{acc[2 * U], acc[2 * U + 1]}), half2v);
        h = __builtin_elementwise_max(h, (half2v{(half_t)0, (half_t)0}));
        yh[e0] = h[0], yh[e0 + 1] = h[1];
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float v = relu(acc[2 * U + r]);
            const half_t h = (half_t)v;
            yh[e0 + r] = h;
            yl[e0 + r] = (half_t)(v - (float)h);
        }
    }
}

// ---- positional encodings straight into B fragments.  Lane half h owns, per sample, 16 of the 32 (sin, cos) pairs
// of the 63-wide point encoding (q = 16h + i; pairs 30, 31 carry the raw coordinates); pair i sits in k-step i>>2,
// elements 2*(i&3) and 2*(i&3)+1.
template <bool SPLIT, bool ACCURATE>
__device__ __forceinline__ void encode_point_w(const double (&p)[3], int h, half8 (&hi)[4], half8 (&lo)[4]) {
    const double rx = p[0] * kInv2Pi, ry = p[1] * kInv2Pi, rz = p[2] * kInv2Pi;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = 16 * h + i;
        const int band = q / 3, a = q - 3 * band;
        const double r = a == 0 ? rx : (a == 1 ? ry : rz);
        float s, c;
        sincos_turns<ACCURATE>(ldexp(r, band), s, c);
        if (i >= 14) {  // only lane half 1 reaches the raw pairs 30, 31
            if (q == 30) s = (float)p[0], c = (float)p[1];
            if (q == 31) s = (float)p[2], c = 0.0f;
        }
        put_pair<SPLIT>(hi[i >> 2], lo[i >> 2], 2 * (i & 3), s, c);
    }
}

// 27-wide direction encoding -> 2 k-steps: lane half h owns pairs q = 8h + i of the 12 (band, coord) pairs; pairs 12,
// 13 carry (x, y), (z, 0); 14, 15 are padding.
template <bool SPLIT, bool ACCURATE>
__device__ __forceinline__ void encode_dir_w(const double (&d)[3], int h, half8 (&hi)[2], half8 (&lo)[2]) {
    const double rx = d[0] * kInv2Pi, ry = d[1] * kInv2Pi, rz = d[2] * kInv2Pi;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = 8 * h + i;
        const int band = q / 3, a = q - 3 * band;
        const double r = a == 0 ? rx : (a == 1 ? ry : rz);
        float s, c;
        sincos_turns<ACCURATE>(ldexp(r, band), s, c);
        if (q == 12) s = (float)d[0], c = (float)d[1];
        if (q == 13) s = (float)d[2], c = 0.0f;
        if (q >= 14) s = 0.0f, c = 0.0f;
        put_pair<SPLIT>(hi[i >> 2], lo[i >> 2], 2 * (i & 3), s, c);
    }
}

// ---- the twelve dense layers of MLP_style (reference models.py:95-117) on one wide weight stream.
//   dir_fn(de_h[2], de_l[2])   fills the direction encoding (called in front of the colour head)
//   sigma_fn(sigma)            sigma of sample lane&31, valid in lanes 0..31
//   remap_fn(ic<rt>, ic<u>, acc)   base_remap accumulator pair (pre-ReLU), may be a no-op
//   rgb_fn(acc)                colour head rows 0..2 BEFORE the sigmoid in acc[0..2], valid in lanes 0..31
template <class C, bool FULL, class WS, class DirFn, class SigmaFn, class RemapFn, class RgbFn>
__device__ __forceinline__ void nerf_chain_w(WS& ws, lds_cptr bias_lane, const half8 (&pe_h)[4], const half8 (&pe_l)[4],
                                             DirFn&& dir_fn, SigmaFn&& sigma_fn, RemapFn&& remap_fn, RgbFn&& rgb_fn) {
    using L = NerfLayoutW;
    half8 Xh[16], Xl[16], Yh[16], Yl[16];
    auto to_Y = [&](auto rt_, auto u_, const float16v& acc) {
        constexpr int rt = decltype(rt_)::value, u = decltype(u_)::value;
        store_act_w<C, u>(acc, Yh[2 * rt + (u >> 2)], Yl[2 * rt + (u >> 2)]);
    };
    auto to_X = [&](auto rt_, auto u_, const float16v& acc) {
        constexpr int rt = decltype(rt_)::value, u = decltype(u_)::value;
        store_act_w<C, u>(acc, Xh[2 * rt + (u >> 2)], Xl[2 * rt + (u >> 2)]);
    };
    dense_layer_w<C, L::frag0(0), 4, 8, L::bias0(0)>(ws, bias_lane, pe_h, pe_l, to_Y);
    dense_layer_w<C, L::frag0(1), 16, 8, L::bias0(1)>(ws, bias_lane, Yh, Yl, to_X);
    dense_layer_w<C, L::frag0(2), 16, 8, L::bias0(2)>(ws, bias_lane, Xh, Xl, to_Y);
    dense_layer_w<C, L::frag0(3), 16, 8, L::bias0(3)>(ws, bias_lane, Yh, Yl, to_X);
    dense_layer_w<C, L::frag0(4), 16, 8, L::bias0(4)>(ws, bias_lane, Xh, Xl, to_Y);
    {
        // skip layer: reference input is cat(pe, h) (models.py:98-99); k order here is [h | pe]
        half8 Bh[20], Bl[20];
#pragma unroll
        for (int k = 0; k < 16; ++k) Bh[k] = Yh[k], Bl[k] = Yl[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) Bh[16 + k] = pe_h[k], Bl[16 + k] = pe_l[k];
        dense_layer_w<C, L::frag0(5), 20, 8, L::bias0(5)>(ws, bias_lane, Bh, Bl, to_X);
    }
    dense_layer_w<C, L::frag0(6), 16, 8, L::bias0(6)>(ws, bias_lane, Xh, Xl, to_Y);
    dense_layer_w<C, L::frag0(7), 16, 8, L::bias0(7)>(ws, bias_lane, Yh, Yl, to_X);
    // sigma head (models.py:103): row 0 of a 32-row tile -> lanes 0..31, register 0
    dense_layer_w<C, L::frag0(8), 16, 1, L::bias0(8)>(ws, bias_lane, Xh, Xl, [&](auto, auto u_, const float16v& acc) {
        if constexpr (decltype(u_)::value == 0) sigma_fn(acc[0]);
    });
    if constexpr (FULL) {
        // base_remap (models.py:106) and the colour head (models.py:107-111)
        dense_layer_w<C, L::frag0(9), 16, 8, L::bias0(9)>(ws, bias_lane, Xh, Xl, [&](auto rt_, auto u_, const float16v& acc) {
            to_Y(rt_, u_, acc);
            remap_fn(rt_, u_, acc);
        });
        half8 Zh[8], Zl[8];
        {
            half8 Bh[18], Bl[18];
            {
                half8 dh[2], dl[2];
                dir_fn(dh, dl);
                Bh[16] = dh[0], Bh[17] = dh[1], Bl[16] = dl[0], Bl[17] = dl[1];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) Bh[k] = Yh[k], Bl[k] = Yl[k];
            dense_layer_w<C, L::frag0(10), 18, 4, L::bias0(10)>(ws, bias_lane, Bh, Bl, [&](auto rt_, auto u_, const float16v& acc) {
                constexpr int rt = decltype(rt_)::value, u = decltype(u_)::value;
                store_act_w<C, u>(acc, Zh[2 * rt + (u >> 2)], Zl[2 * rt + (u >> 2)]);
            });
        }
        dense_layer_w<C, L::frag0(11), 8, 1, L::bias0(11)>(ws, bias_lane, Zh, Zl, [&](auto, auto u_, const float16v& acc) {
            if constexpr (decltype(u_)::value == 1) rgb_fn(acc);
        });
    }
}

}  // namespace tgtc
