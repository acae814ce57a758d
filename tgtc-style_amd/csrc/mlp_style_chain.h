// The stylised per-tile chain on B fragments in registers (reference rendering.py:122-142, models.py:120-180): stream map of
// the three packed nets, the scratch slab helpers, the concat MLP and the style MLP's layers 1..7.  Shared by the per-sample
// kernels (mlp_style.hip) and the stylised ray kernel (render_styled_fused.hip).
#pragma once
#include "mlp_core.h"
#include "mlp_layouts.h"

namespace tgtc {

constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }
constexpr int kTrunkFrags = nerf_frag0(10);  // L0..L7 + sigma + remap = 1096

template <class C>
struct StyledMap {
    static constexpr int F_CONCAT = 0;
    static constexpr int F_NERF = kConcatFrags;  // 656: a whole number of chunks in both modes
    static constexpr int NERF_SPAN = round_up(kTrunkFrags, C::FPC);
    static constexpr int GAP = NERF_SPAN - kTrunkFrags;  // fragments of the NeRF stream's colour head we fly over
    static constexpr int F_STYLE = F_NERF + NERF_SPAN;
    static constexpr int NFRAG = F_STYLE + kStyleFrags;
    static constexpr int NSEG = 3;
    static constexpr int chunk0(int i) {
        return i == 0 ? 0 : i == 1 ? F_NERF / C::FPC : i == 2 ? F_STYLE / C::FPC : (1 << 30);
    }
    static_assert(kConcatFrags % C::FPC == 0, "concat stream must end on a chunk boundary");
};

constexpr int kStashBytesPerWG = 131072;  // 8 k-steps x NCT x (hi[,lo]) x 256 lanes x 16 B in both modes


template <class C, int KSN>
__device__ __forceinline__ void append(half8 (&Bh)[KSN][C::NCT], half8 (&Bl)[KSN][C::NCT], int at,
                                       const half8 (&h)[C::NCT], const half8 (&l)[C::NCT]) {
#pragma unroll
    for (int c = 0; c < C::NCT; ++c) Bh[at][c] = h[c], Bl[at][c] = l[c];
}

// The slab pointer handed to these helpers is the lane's own (slab + tid*16), laundered through an empty asm
// at every use: the slab addresses are invariant across the persistent tile loop, and without this hipcc hoists
// all 2 x 32 address pairs out of the loop and spills hundreds of registers to keep them alive.
__device__ __forceinline__ char* launder(char* p) {
    asm volatile("" : "+v"(p));
    return p;
}
template <class C>
__device__ __forceinline__ void stash_store(char* lane_slab, int ks, int c, half8 h, half8 l) {
    constexpr int P = C::SPLIT ? 2 : 1;
    if constexpr (kAbl & 16) {   // timing ablation: no slab traffic (the values are consumed, results are garbage)
        asm volatile("" ::"v"(h), "v"(l));
        return;
    }
    half8* p = reinterpret_cast<half8*>(launder(lane_slab)) + (size_t)((ks * C::NCT + c) * P) * (C::NWAVES * 64);
    p[0] = h;
    if constexpr (C::SPLIT) p[C::NWAVES * 64] = l;
}
template <class C>
__device__ __forceinline__ void stash_load(char* lane_slab, half8 (&Ah)[8][C::NCT], half8 (&Al)[8][C::NCT]) {
    constexpr int P = C::SPLIT ? 2 : 1;
    if constexpr (kAbl & 16) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int c = 0; c < C::NCT; ++c) asm volatile("" : "=v"(Ah[ks][c]), "=v"(Al[ks][c]));
        return;
    }
    const half8* base = reinterpret_cast<const half8*>(launder(lane_slab));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int c = 0; c < C::NCT; ++c) {
            const half8* p = base + (size_t)((ks * C::NCT + c) * P) * (C::NWAVES * 64);
            Ah[ks][c] = p[0];
            if constexpr (C::SPLIT) Al[ks][c] = p[C::NWAVES * 64];
        }
}

// ------------------------------------------------------------------------------------------------
// the concat MLP on B fragments already in registers; result (256 features) lands in Yh/Yl
template <class C, int F0, int B0, class WS>
__device__ __forceinline__ void concat_mlp(WS& ws, lds_cptr bias_lane, const half8 (&pe_h)[2][C::NCT],
                                           const half8 (&pe_l)[2][C::NCT], const half8 (&z_h)[C::NCT],
                                           const half8 (&z_l)[C::NCT], half8 (&Xh)[8][C::NCT], half8 (&Xl)[8][C::NCT],
                                           half8 (&Yh)[8][C::NCT], half8 (&Yl)[8][C::NCT]) {
    constexpr int NCT = C::NCT;
    auto to_Y = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][c], Yl[rt / 2][c]);
    };
    auto to_X = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][c], Xl[rt / 2][c]);
    };
    {
        half8 Bh[3][NCT], Bl[3][NCT];
        append<C>(Bh, Bl, 0, pe_h[0], pe_l[0]);
        append<C>(Bh, Bl, 1, pe_h[1], pe_l[1]);
        append<C>(Bh, Bl, 2, z_h, z_l);
        dense_layer<C, F0 + concat_frag0(0), 3, 16, B0 + 0>(ws, bias_lane, Bh, Bl, to_Y);
    }
    auto hidden = [&](auto layer_, const half8 (&Ah)[8][NCT], const half8 (&Al)[8][NCT], auto&& epi) {
        constexpr int l = decltype(layer_)::value;
        half8 Bh[9][NCT], Bl[9][NCT];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Ah[k], Al[k]);
        append<C>(Bh, Bl, 8, z_h, z_l);
        dense_layer<C, F0 + concat_frag0(l), 9, 16, B0 + 256 * l>(ws, bias_lane, Bh, Bl, epi);
    };
    hidden(ic<1>{}, Yh, Yl, to_X);
    hidden(ic<2>{}, Xh, Xl, to_Y);
    hidden(ic<3>{}, Yh, Yl, to_X);
    {
        half8 Bh[11][NCT], Bl[11][NCT];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Xh[k], Xl[k]);
        append<C>(Bh, Bl, 8, z_h, z_l);
        append<C>(Bh, Bl, 9, pe_h[0], pe_l[0]);
        append<C>(Bh, Bl, 10, pe_h[1], pe_l[1]);
        dense_layer<C, F0 + concat_frag0(4), 11, 16, B0 + 256 * 4>(ws, bias_lane, Bh, Bl, to_Y);
    }
}

// style MLP layers 1..7 (input in Xh/Xl, the 256 outputs of layer 0); emit(c, acc) receives the rgb tile
template <class C, int F0, int B0, class WS, class Emit>
__device__ __forceinline__ void style_tail(WS& ws, lds_cptr bias_lane, const half8 (&pe_h)[2][C::NCT],
                                           const half8 (&pe_l)[2][C::NCT], const half8 (&z_h)[C::NCT],
                                           const half8 (&z_l)[C::NCT], half8 (&Xh)[8][C::NCT], half8 (&Xl)[8][C::NCT],
                                           half8 (&Yh)[8][C::NCT], half8 (&Yl)[8][C::NCT], Emit&& emit) {
    constexpr int NCT = C::NCT;
    auto to_Y = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][c], Yl[rt / 2][c]);
    };
    auto to_X = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][c], Xl[rt / 2][c]);
    };
    auto hidden = [&](auto layer_, const half8 (&Ah)[8][NCT], const half8 (&Al)[8][NCT], auto&& epi) {
        constexpr int l = decltype(layer_)::value;
        half8 Bh[9][NCT], Bl[9][NCT];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Ah[k], Al[k]);
        append<C>(Bh, Bl, 8, z_h, z_l);
        dense_layer<C, F0 + style_frag0(l), 9, kStyleRT[l], B0 + style_bias0(l)>(ws, bias_lane, Bh, Bl, epi);
    };
    hidden(ic<1>{}, Xh, Xl, to_Y);
    hidden(ic<2>{}, Yh, Yl, to_X);
    hidden(ic<3>{}, Xh, Xl, to_Y);
    {
        half8 Bh[11][NCT], Bl[11][NCT];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Yh[k], Yl[k]);
        append<C>(Bh, Bl, 8, z_h, z_l);
        append<C>(Bh, Bl, 9, pe_h[0], pe_l[0]);
        append<C>(Bh, Bl, 10, pe_h[1], pe_l[1]);
        dense_layer<C, F0 + style_frag0(4), 11, 16, B0 + style_bias0(4)>(ws, bias_lane, Bh, Bl, to_X);
    }
    hidden(ic<5>{}, Xh, Xl, to_Y);
    hidden(ic<6>{}, Yh, Yl, to_X);
    hidden(ic<7>{}, Xh, Xl, [&](auto, auto c_, auto h_, const float4v& acc) { emit(c_, h_, acc); });
}

template <bool SPLIT>
__device__ __forceinline__ void splat8(float v, half8& hi, half8& lo) {
    const half_t h = (half_t)v;
    const half_t l = (half_t)(v - (float)h);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = h;
        if constexpr (SPLIT) lo[j] = l;
    }
}

}  // namespace tgtc
