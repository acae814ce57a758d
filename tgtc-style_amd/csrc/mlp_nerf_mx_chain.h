// The twelve dense layers of MLP_style (reference models.py:95-117 inside StyleNerf.forward :216-223) in
// TGTC_PREC_FP16_FP6 (mlp_mx.h), one column tile per wave.  Shared by the per-sample kernel (mlp_nerf_mx.hip) and
// the fused ray kernel (render_fused.hip).  Callbacks as in mlp_nerf_chain.h, without the column-tile index:
//   dir_fn(de_h, de_l), sigma_fn(sigma), remap_fn(ic<rt>, ic<half>, acc), rgb_fn(ic<half>, acc)
#pragma once
#include "mlp_layouts.h"
#include "mlp_mx.h"
#include "mx_asm.h"

// The generated instruction streams of tools/gen_mx_asm.py (mx_asm_nerf.inc; profiles/r4_kernel_variants.md) in the fused ray
// kernel's FULL fp16mx pass.  2 (shipped): the whole pass -- all twelve layers -- is ONE stream (render_fused.h); 1: only the
// 256 -> 256 layers 1-4 and 6-7, as blocks between HIP layers (measured 1.5 % slower than 0: every HIP <-> asm transition
// drains the LDS queue); 0: the HIP loop (dense_mx) everywhere.  The per-sample kernel always runs the HIP loop.
#ifndef TGTC_MX_ASM
#define TGTC_MX_ASM 2
#endif

namespace tgtc {

#include "mx_asm_nerf.inc"

constexpr int mx_bytes_upto(const MxTable& t, int nq) {
    const int end = t.off[nq - 1] + (t.npe[nq - 1] ? t.npe[nq - 1] * 2048 : kMxKGroupBytes);
    return (end + kChunkBytes - 1) / kChunkBytes * kChunkBytes;
}
constexpr int nerf_mx_groups(bool full) { return full ? kNerfMxTable.first[12] : kNerfMxTable.first[9]; }
constexpr int nerf_mx_units(bool full) { return mx_bytes_upto(kNerfMxTable, nerf_mx_groups(full)) / 1024; }

// smem / wave (optional): the workgroup's LDS base and the wave index; given, the persistent FULL pass runs its 256 -> 256
// layers 1-4 and 6-7 as the generated instruction streams (TGTC_MX_ASM) and re-derives the per-lane addresses around them.
template <class C, bool FULL, class Reader, class DirFn, class SigmaFn, class RemapFn, class RgbFn>
__device__ __forceinline__ void nerf_chain_mx(Reader& rd, lds_cptr bias_lane_in, lds_cptr rs_lane_in, const half8 (&Ph)[2],
                                              const half8 (&Pl)[2], DirFn&& dir_fn, SigmaFn&& sigma_fn,
                                              RemapFn&& remap_fn, RgbFn&& rgb_fn, char* smem = nullptr, int wave = 0) {
    lds_cptr bias_lane = bias_lane_in, rs_lane = rs_lane_in;
    auto relane = [&] {
        rd.relane(smem, wave);
        const int lane = fresh_lane_id();
        bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * (lane >> 4));
        rs_lane = opaque((lds_cptr)smem + C::RING_BYTES + kNerfMxScaleOff + 2 * (lane & 15));
    };
    using L = NerfLayout;
    constexpr int NQ = nerf_mx_groups(FULL);
    constexpr const MxTable& T = kNerfMxTable;
    MxAct<2> X, Y;
    MxAct<1> none;  // layers without an activation input
    half8 l16[4];
    const half8 nop[1] = {};
    auto to_Y = [&](auto rt_, auto h_, const float4v& acc) { mx_store_act<decltype(rt_)::value, decltype(h_)::value>(acc, Y, l16); };
    auto to_X = [&](auto rt_, auto h_, const float4v& acc) { mx_store_act<decltype(rt_)::value, decltype(h_)::value>(acc, X, l16); };

    // the generated streams hand over a DEPTH-1 persistent reader with 16 KiB chunks (tools/gen_mx_asm.py); every other
    // reader (the per-sample kernel's draining ring, development depths) keeps the HIP loop
    const bool have_lds_base = smem != nullptr;
    (void)have_lds_base;
    constexpr bool ASM = TGTC_MX_ASM && FULL && Reader::Ring::IS_PERSIST && Reader::DEPTH == 1 && Reader::Ring::STAG == 0 &&
                         kChunkBytes == 16384 && C::SLOTS == 8;
    dense_mx<C, T.first[0], NQ, 16, 0, 2, L::bias0(0)>(rd, bias_lane, rs_lane, none, Ph, Pl, to_Y);
    // the skip layer's and the colour head's encodings, pinned across the asm blocks (mx_asm_nerf.inc): with the streams the
    // direction is encoded up front -- encoded late it is hoisted by hipcc anyway, spilled, and reloaded behind a vmcnt(0)
    half8 pe_keep[6] = {Ph[0], Ph[1], Pl[0], Pl[1], half8{}, half8{}};
    if constexpr (ASM) {
        dir_fn(pe_keep[4], pe_keep[5]);
        relane();
        mx_asm_nerf_full_l1_4(rd, bias_lane, rs_lane, Y, Y, pe_keep);
        relane();
    } else {
        dense_mx<C, T.first[1], NQ, 16, 2, 0, L::bias0(1)>(rd, bias_lane, rs_lane, Y, nop, nop, to_X);
        dense_mx<C, T.first[2], NQ, 16, 2, 0, L::bias0(2)>(rd, bias_lane, rs_lane, X, nop, nop, to_Y);
        dense_mx<C, T.first[3], NQ, 16, 2, 0, L::bias0(3)>(rd, bias_lane, rs_lane, Y, nop, nop, to_X);
        dense_mx<C, T.first[4], NQ, 16, 2, 0, L::bias0(4)>(rd, bias_lane, rs_lane, X, nop, nop, to_Y);
    }
    // skip layer: reference input is cat(pe, h) (models.py:98-99); k order here is [h | pe]
    {
        const half8 Ph5[2] = {pe_keep[0], pe_keep[1]}, Pl5[2] = {pe_keep[2], pe_keep[3]};
        dense_mx<C, T.first[5], NQ, 16, 2, 2, L::bias0(5)>(rd, bias_lane, rs_lane, Y, Ph5, Pl5, to_X);
    }
    if constexpr (ASM) {
        relane();
        mx_asm_nerf_full_l6_7(rd, bias_lane, rs_lane, X, X, pe_keep);
        relane();
    } else {
        dense_mx<C, T.first[6], NQ, 16, 2, 0, L::bias0(6)>(rd, bias_lane, rs_lane, X, nop, nop, to_Y);
        dense_mx<C, T.first[7], NQ, 16, 2, 0, L::bias0(7)>(rd, bias_lane, rs_lane, Y, nop, nop, to_X);
    }

    // sigma head (models.py:103): row 0 of a 16-row tile -> lanes 0..15, register 0
    dense_mx<C, T.first[8], NQ, 1, 2, 0, L::bias0(8)>(rd, bias_lane, rs_lane, X, nop, nop, [&](auto, auto h_, const float4v& acc) {
        if constexpr (decltype(h_)::value == 0) sigma_fn(acc[0]);
    });

    if constexpr (FULL) {
        // base_remap (models.py:106) and the colour head (models.py:107-111)
        dense_mx<C, T.first[9], NQ, 16, 2, 0, L::bias0(9)>(rd, bias_lane, rs_lane, X, nop, nop, [&](auto rt_, auto h_, const float4v& acc) {
            mx_store_act<decltype(rt_)::value, decltype(h_)::value>(acc, Y, l16);
            remap_fn(rt_, h_, acc);
        });
        MxAct<1> Z;
        half8 Dh[1], Dl[1];
        if constexpr (ASM) Dh[0] = pe_keep[4], Dl[0] = pe_keep[5];
        else dir_fn(Dh[0], Dl[0]);
        dense_mx<C, T.first[10], NQ, 8, 2, 1, L::bias0(10)>(rd, bias_lane, rs_lane, Y, Dh, Dl, [&](auto rt_, auto h_, const float4v& acc) {
            mx_store_act<decltype(rt_)::value, decltype(h_)::value>(acc, Z, l16);
        });
        dense_mx<C, T.first[11], NQ, 1, 1, 0, L::bias0(11)>(rd, bias_lane, rs_lane, Z, nop, nop,
                                                             [&](auto, auto h_, const float4v& acc) { rgb_fn(h_, acc); });
    }
}

}  // namespace tgtc
