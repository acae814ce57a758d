// The twelve dense layers of MLP_style (reference models.py:95-117, inside StyleNerf.forward :216-223) on one
// weight stream, for the fp16 / fp16x3 kernels: activations ping-pong between two register sets X / Y and never
// leave registers (mlp_core.h).  Shared by the per-sample kernels (mlp_nerf.hip) and the fused ray kernel
// (render_fused.hip); the callers differ in where the encodings come from and where sigma / rgb go:
//   dir_fn(ic<c>, de_h, de_l)      fills the direction encoding of column tile c (called in front of the colour head)
//   sigma_fn(ic<c>, sigma)         sigma of sample lane&15 of column tile c, valid in lanes 0..15
//   remap_fn(ic<rt>, ic<c>, ic<half>, acc)   base_remap rows (pre-ReLU accumulator half), may be a no-op
//   rgb_fn(ic<c>, ic<half>, acc)   colour head rows 0..2 BEFORE the sigmoid, valid in lanes 0..15
#pragma once
#include "mlp_core.h"
#include "mlp_layouts.h"

namespace tgtc {

template <class C, bool FULL, class WS, class DirFn, class SigmaFn, class RemapFn, class RgbFn>
__device__ __forceinline__ void nerf_chain(WS& ws, lds_cptr bias_lane, const half8 (&pe_h)[2][C::NCT],
                                           const half8 (&pe_l)[2][C::NCT], DirFn&& dir_fn, SigmaFn&& sigma_fn,
                                           RemapFn&& remap_fn, RgbFn&& rgb_fn) {
    constexpr int NCT = C::NCT;
    using L = NerfLayout;
    half8 Xh[8][NCT], Xl[8][NCT], Yh[8][NCT], Yl[8][NCT];
    auto to_Y = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][c], Yl[rt / 2][c]);
    };
    auto to_X = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][c], Xl[rt / 2][c]);
    };

    dense_layer<C, L::frag0(0), 2, 16, L::bias0(0)>(ws, bias_lane, pe_h, pe_l, to_Y);
    dense_layer<C, L::frag0(1), 8, 16, L::bias0(1)>(ws, bias_lane, Yh, Yl, to_X);
    dense_layer<C, L::frag0(2), 8, 16, L::bias0(2)>(ws, bias_lane, Xh, Xl, to_Y);
    dense_layer<C, L::frag0(3), 8, 16, L::bias0(3)>(ws, bias_lane, Yh, Yl, to_X);
    dense_layer<C, L::frag0(4), 8, 16, L::bias0(4)>(ws, bias_lane, Xh, Xl, to_Y);
    {
        // skip layer: reference input is cat(pe, h) (models.py:98-99); k order here is [h | pe]
        half8 Bh[10][NCT], Bl[10][NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
#pragma unroll
            for (int k = 0; k < 8; ++k) Bh[k][c] = Yh[k][c], Bl[k][c] = Yl[k][c];
            Bh[8][c] = pe_h[0][c], Bh[9][c] = pe_h[1][c], Bl[8][c] = pe_l[0][c], Bl[9][c] = pe_l[1][c];
        }
        dense_layer<C, L::frag0(5), 10, 16, L::bias0(5)>(ws, bias_lane, Bh, Bl, to_X);
    }
    dense_layer<C, L::frag0(6), 8, 16, L::bias0(6)>(ws, bias_lane, Xh, Xl, to_Y);
    dense_layer<C, L::frag0(7), 8, 16, L::bias0(7)>(ws, bias_lane, Yh, Yl, to_X);

    // sigma head (models.py:103): row 0 of a 16-row tile -> lanes 0..15, register 0
    dense_layer<C, L::frag0(8), 8, 1, L::bias0(8)>(ws, bias_lane, Xh, Xl, [&](auto, auto c_, auto h_, const float4v& acc) {
        if constexpr (decltype(h_)::value == 0) sigma_fn(c_, acc[0]);
    });

    if constexpr (FULL) {
        // base_remap (models.py:106) and the colour head (models.py:107-111)
        dense_layer<C, L::frag0(9), 8, 16, L::bias0(9)>(ws, bias_lane, Xh, Xl, [&](auto rt_, auto c_, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
            store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][c], Yl[rt / 2][c]);
            remap_fn(rt_, c_, h_, acc);
        });
        half8 Zh[4][NCT], Zl[4][NCT];
        {
            half8 Bh[9][NCT], Bl[9][NCT];
            static_for<NCT>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                dir_fn(c_, Bh[8][c], Bl[8][c]);
#pragma unroll
                for (int k = 0; k < 8; ++k) Bh[k][c] = Yh[k][c], Bl[k][c] = Yl[k][c];
            });
            dense_layer<C, L::frag0(10), 9, 8, L::bias0(10)>(ws, bias_lane, Bh, Bl, [&](auto rt_, auto c_, auto h_, const float4v& acc) {
                constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
                store_act<C, rt, decltype(h_)::value>(acc, Zh[rt / 2][c], Zl[rt / 2][c]);
            });
        }
        dense_layer<C, L::frag0(11), 4, 1, L::bias0(11)>(ws, bias_lane, Zh, Zl,
                                                          [&](auto, auto c_, auto h_, const float4v& acc) { rgb_fn(c_, h_, acc); });
    }
}

}  // namespace tgtc
