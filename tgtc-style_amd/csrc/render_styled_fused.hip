// The whole STYLISED render of a ray in one persistent kernel (reference rendering.py:109-182 render_style;
// models.py:95-117, 120-180; utils.py:354-386, 509-531, 573-609) -- render_fused.hip with the stylised chain as its fine pass:
//
//   coarse depths -> [PE + coarse NeRF MLP (sigma)] x Nc/16 tiles -> weights -> inverse-CDF fine sampling + merge
//   -> [PE + concat MLP + fine NeRF trunk (sigma, base_remap) + style MLP (rgb)] x (Nc+Nf)/16 tiles -> alpha compositing
//
// A wavefront owns a ray, as there: depths, weights and the compositing state sit in the wave's LDS strip and registers, the
// ray's latent (32 floats, rendering.py:126) in the strip too, and HBM sees a ray and a latent in and a pixel out.  No
// per-sample tensor exists: the 1.1 GB workspace of the per-sample chain (render.hip) is not touched.  The eight waves walk
// the coarse stream and the stylised pass's three streams (concat | NeRF trunk | style, mlp_style_chain.h StyledMap) through
// ONE ring that never drains; the third activation set of the stylised chain is parked in the style handle's per-workgroup
// slab exactly as in the per-sample kernel (mlp_style.hip; L2 resident, 2.9 % of the frame).
//
// LDS: the stylised pass needs two bias tables at once (NeRF 16 KiB + concat/style 16 KiB), so the ring gives up one of its
// eight 16 KiB slots: 7 x 16 + 16 + 16 + 8 x 1.9 KiB of strips = 159.25 KiB.
//
// Per-tile arithmetic is the code of the per-sample kernels (mlp_nerf_chain.h, mlp_style_chain.h, raymarch_wave.h): a ray's
// result does not depend on the wave, workgroup or launch that renders it, and equals the chain's to rounding.
#include "render_fused.h"
#include "mlp_style_chain.h"

namespace tgtc {


constexpr int kStyledSlots = kRingSlots - 1;
constexpr int kStripZ = kStripAcc + 8;                                  // 32 floats of latent behind the compositing state
constexpr int kStyledStripBytes = kFusedStripBytes + 32 * 4;

// One stylised pass over this wave's tile (depth tt) of its ray: *sig_slot (this lane's word of the strip) / col valid in lanes 0..15.
template <class C>
__device__ __forceinline__ void styled_pass(char* smem, int wave, int lane, const FusedStyledArgs& a, const char* next_stream,
                                            const double (&o)[3], const double (&d)[3], float tt, const float* z32, char* lane_slab,
                                            float* sig_slot, float (&col)[3]) {
    static_assert(C::NCT == 1 && C::SPLIT, "the stylised ray kernel is built for fp16x3 (one column tile per wave)");
    using Map = StyledMap<C>;
    using L = NerfLayout;
    using StreamT = WeightStream<C, Map, true, TGTC_FUSED_X3_ASM_DMA>;
    const int g = lane >> 4;
    half8 pe_h[2][1], pe_l[2][1], z_h[1], z_l[1], zb_h[1], zb_l[1];
    {
        double p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = o[k] + (double)tt * d[k];   // rendering.py:118 / utils.py:529,578
        half8 h2[2], l2[2];
        encode_point<true, true>(p, g, h2, l2, nullptr);
        pe_h[0][0] = h2[0], pe_h[1][0] = h2[1], pe_l[0][0] = l2[0], pe_l[1][0] = l2[1];
        // the latent, and its mean over the 32 channels broadcast back to 32 (rendering.py:126, :139)
        load_vec32<true>(z32, g, z_h[0], z_l[0]);
        float zs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) zs += z32[8 * g + j];
        zs += __shfl_xor(zs, 16);
        zs += __shfl_xor(zs, 32);
        splat8<true>(zs * (1.0f / 32.0f), zb_h[0], zb_l[0]);
    }
    const lds_cptr nerf_bias = opaque((lds_cptr)smem + C::RING_BYTES + 16 * g);
    const lds_cptr pair_bias = opaque((lds_cptr)smem + C::RING_BYTES + kNerfBiasBytes + 16 * g);
    StreamT ws;
    const char* const streams[3] = {a.concat_stream, a.ray.net_f + kNerfBiasBytes, a.style_stream};
    ws.init(streams, smem, wave, lane);
    const char* const next_src = StreamT::lane_src(next_stream, wave, lane);
    ws.next = ws.src[0];   // the stream this pass enters (its first chunks are in flight) ...
    ws.enter();
    ws.next = next_src;    // ... and the one its look-ahead runs into

    half8 Xh[8][1], Xl[8][1], Yh[8][1], Yl[8][1];
    // ---- concat MLP -> Y, parked in the slab while the trunk runs (mlp_style.hip)
    concat_mlp<C, Map::F_CONCAT, 0>(ws, pair_bias, pe_h, pe_l, z_h, z_l, Xh, Xl, Yh, Yl);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) stash_store<C>(lane_slab, ks, 0, Yh[ks][0], Yl[ks][0]);

    // ---- NeRF trunk (models.py:95-101)
    auto to_Y = [&](auto rt_, auto, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][0], Yl[rt / 2][0]);
    };
    auto to_X = [&](auto rt_, auto, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][0], Xl[rt / 2][0]);
    };
    constexpr int FN = Map::F_NERF;
    dense_layer<C, FN + L::frag0(0), 2, 16, L::bias0(0)>(ws, nerf_bias, pe_h, pe_l, to_Y);
    dense_layer<C, FN + L::frag0(1), 8, 16, L::bias0(1)>(ws, nerf_bias, Yh, Yl, to_X);
    dense_layer<C, FN + L::frag0(2), 8, 16, L::bias0(2)>(ws, nerf_bias, Xh, Xl, to_Y);
    dense_layer<C, FN + L::frag0(3), 8, 16, L::bias0(3)>(ws, nerf_bias, Yh, Yl, to_X);
    dense_layer<C, FN + L::frag0(4), 8, 16, L::bias0(4)>(ws, nerf_bias, Xh, Xl, to_Y);
    {
        half8 Bh[10][1], Bl[10][1];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Yh[k], Yl[k]);
        append<C>(Bh, Bl, 8, pe_h[0], pe_l[0]);
        append<C>(Bh, Bl, 9, pe_h[1], pe_l[1]);
        dense_layer<C, FN + L::frag0(5), 10, 16, L::bias0(5)>(ws, nerf_bias, Bh, Bl, to_X);
    }
    dense_layer<C, FN + L::frag0(6), 8, 16, L::bias0(6)>(ws, nerf_bias, Xh, Xl, to_Y);
    dense_layer<C, FN + L::frag0(7), 8, 16, L::bias0(7)>(ws, nerf_bias, Yh, Yl, to_X);
    dense_layer<C, FN + L::frag0(8), 8, 1, L::bias0(8)>(ws, nerf_bias, Xh, Xl, [&](auto, auto, auto h_, const float4v& acc) {
        // models.py:103.  sigma waits for the colour in the wave's strip, not in a register: it would live through the 1 200
        // fragments of the style MLP, and the compiler spills the whole accumulator tile for it (a scratch store in the loop)
        if constexpr (decltype(h_)::value == 0) *sig_slot = acc[0];
    });
    dense_layer<C, FN + L::frag0(9), 8, 16, L::bias0(9)>(ws, nerf_bias, Xh, Xl, to_Y);  // base_remap -> Y
    ws.template skip<FN + kTrunkFrags, Map::GAP>();

    // ---- style layer 0 on [remap (Y) | concat_features (slab -> X) | pe | mean z]; outputs stream to the slab
    stash_load<C>(lane_slab, Xh, Xl);
    {
        half8 Bh[19][1], Bl[19][1];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Yh[k], Yl[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, 8 + k, Xh[k], Xl[k]);
        append<C>(Bh, Bl, 16, pe_h[0], pe_l[0]);
        append<C>(Bh, Bl, 17, pe_h[1], pe_l[1]);
        append<C>(Bh, Bl, 18, zb_h, zb_l);
        half8 Th[1], Tl[1];
        dense_layer<C, Map::F_STYLE + style_frag0(0), 19, 16, kConcatBiasFloats + style_bias0(0)>(
            ws, pair_bias, Bh, Bl, [&](auto rt_, auto, auto h_, const float4v& acc) {
                constexpr int rt = decltype(rt_)::value, hf = decltype(h_)::value;
                store_act<C, rt, hf>(acc, Th[0], Tl[0]);
                if constexpr ((rt & 1) && hf == 1) stash_store<C>(lane_slab, rt / 2, 0, Th[0], Tl[0]);
            });
    }
    stash_load<C>(lane_slab, Xh, Xl);
    // ---- style layers 1..7 -> rgb (models.py:172-179)
    style_tail<C, Map::F_STYLE, kConcatBiasFloats>(ws, pair_bias, pe_h, pe_l, zb_h, zb_l, Xh, Xl, Yh, Yl,
                                                   [&](auto, auto h_, const float4v& acc) {
                                                       constexpr int hf = decltype(h_)::value;
#pragma unroll
                                                       for (int r = 2 * hf; r < (hf ? 3 : 2); ++r) col[r] = 1.0f / (1.0f + expf(-acc[r]));
                                                   });
    ws.template finish<StreamT::NCHUNK>();
}

template <int PC>
__global__ void __launch_bounds__(512, 2) fused_styled_kernel(FusedStyledArgs sa) {
    constexpr int PF = TGTC_PREC_FP16X3;
    using CC = typename FusedCfg<PC, kStyledSlots>::C;
    using CF = typename FusedCfg<PF, kStyledSlots>::C;
    static_assert(CC::NWAVES == 8 && CF::NWAVES == 8 && CC::SLOTS == CF::SLOTS, "one ring, eight waves");
    constexpr int NW = 8;
    const FusedArgs& a = sa.ray;

    // ring | NeRF bias table of the running phase | concat + style bias table | per-wave strips
    __shared__ __attribute__((aligned(16))) char smem[CC::RING_BYTES + kNerfBiasBytes + kStylePairBiasBytes + NW * kStyledStripBytes];
    static_assert(sizeof(smem) <= 160 * 1024, "LDS");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto fresh_lane = [&] {   // render_fused.hip: lane-derived addresses must not be hoisted out of the ray loop
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    auto strip = [&]() -> float* {
        return reinterpret_cast<float*>(smem + CC::RING_BYTES + kNerfBiasBytes + kStylePairBiasBytes + wave * kStyledStripBytes);
    };
    auto load_bias = [&](const char* net) {   // NeRF table of a phase: everyone is done with the old one; drained; published by the next enter()
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < kNerfBiasBytes / (NW * 1024); ++j)
            lds_dma16_t<false>(net + (j * NW + wave) * 1024 + lane * 16, smem + CC::RING_BYTES + (j * NW + wave) * 1024);
        wait_vmcnt<0>();
    };
    auto put_acc = [&](const RayAccum& acc) {
        float* s = strip() + kStripAcc;
        if (fresh_lane() == 0) {
            const unsigned long long tb = __builtin_bit_cast(unsigned long long, acc.trans);
            s[0] = __builtin_bit_cast(float, (unsigned)tb), s[1] = __builtin_bit_cast(float, (unsigned)(tb >> 32));
            s[2] = acc.r, s[3] = acc.g, s[4] = acc.b, s[5] = acc.t;
        }
        wave_sync();
    };
    auto get_acc = [&]() {
        wave_sync();
        const float* s = strip() + kStripAcc;
        RayAccum acc;
        acc.trans = __builtin_bit_cast(double, ((unsigned long long)__builtin_bit_cast(unsigned, s[1]) << 32) |
                                                   __builtin_bit_cast(unsigned, s[0]));
        acc.r = s[2], acc.g = s[3], acc.b = s[4], acc.t = s[5];
        return acc;
    };

    // the concat / style bias table never changes: once per kernel (visible after the first ring barrier)
#pragma unroll
    for (int j = 0; j < kStylePairBiasBytes / (NW * 1024); ++j)
        lds_dma16_t<false>(sa.pair_bias + (j * NW + wave) * 1024 + lane * 16,
                           smem + CC::RING_BYTES + kNerfBiasBytes + (j * NW + wave) * 1024);
    {   // chunks 0 .. SLOTS-2 of the first coarse pass: the state every enter() expects
        typename FusedStream<PC, false, kStyledSlots>::type first;
        const char* const one[1] = {a.net_c + kNerfBiasBytes};
        first.init(one, smem, wave, lane);
        first.next = first.src[0];
        first.persist_prologue();
    }
    // this lane's 16-byte column of the workgroup's slab, re-derived per pass (a 64-bit per-lane pointer kept across the
    // passes would be spilled, render_fused.hip)
    auto lane_slab = [&]() -> char* { return sa.slab + (size_t)blockIdx.x * kStashBytesPerWG + (size_t)(wave * 64 + fresh_lane()) * 16; };

    const int tiles_c = a.NC / 16, tiles_f = (a.NC + a.NF) / 16, NT = a.NC + a.NF;
    const long long groups = (a.R + NW - 1) / NW;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long ray_raw = grp * NW + wave;
        const bool valid = ray_raw < a.R;
        const long long ray = valid ? ray_raw : a.R - 1;   // tail: duplicate the last ray, the store is masked
        double o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = uniform_f64(a.rays_o[ray * 3 + k]), d[k] = uniform_f64(a.rays_d[ray * 3 + k]);
        if (a.jitter) {
            float* s_w = strip() + kStripW;
            for (int i = fresh_lane(); i < a.NC; i += 64) s_w[i] = a.jitter[ray * a.NC + i];
        }
        if (fresh_lane() < 32) strip()[kStripZ + fresh_lane()] = sa.z[ray * 32 + fresh_lane()];
        load_bias(a.net_c);

        // ---------------------------------------------------------------- coarse passes (sigma only): render_fused.hip
        put_acc(RayAccum{1.0, 0.f, 0.f, 0.f, 0.f});
        for (int tile = 0; tile < tiles_c; tile += CC::NCT) {
            auto depths = [&](float (&tt)[CC::NCT], float (&tn)[CC::NCT]) {
                const float* s_w = strip() + kStripW;
                const int n = fresh_lane() & 15;
#pragma unroll
                for (int c = 0; c < CC::NCT; ++c) {
                    const int i = 16 * (tile + c) + n, i1 = min(i + 1, a.NC - 1);
                    if (a.jitter) {
                        tt[c] = coarse_t_jittered(i, a.NC, a.near_, a.far_, s_w[i]);
                        tn[c] = coarse_t_jittered(i1, a.NC, a.near_, a.far_, s_w[i1]);
                    } else {
                        tt[c] = coarse_t(i, a.NC, a.near_, a.far_);
                        tn[c] = coarse_t(i1, a.NC, a.near_, a.far_);
                    }
                }
            };
            float tt[CC::NCT], tn[CC::NCT], sig[CC::NCT], col[CC::NCT][3];
            wave_sync();
            depths(tt, tn);
            const bool last = tile + CC::NCT >= tiles_c;
            fused_pass<PC, false, kStyledSlots>(smem, wave, fresh_lane(), a.net_c, last ? sa.concat_stream : a.net_c + kNerfBiasBytes, o, d, tt, sig, col);
            depths(tt, tn);
            RayAccum acc = get_acc();
            float w[CC::NCT];
            const int ln = fresh_lane(), n = ln & 15, g = ln >> 4;
#pragma unroll
            for (int c = 0; c < CC::NCT; ++c) {
                const int i = 16 * (tile + c) + n;
                w[c] = composite_tile<false>(sig[c], 0.f, 0.f, 0.f, tt[c], (i + 1 < a.NC) ? tn[c] - tt[c] : 1e10f, acc);
            }
            wave_sync();
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < CC::NCT; ++c) {
                    const int i = 16 * (tile + c) + n;
                    strip()[kStripAll + i] = tt[c], strip()[kStripW + i] = w[c];
                }
            }
            put_acc(acc);
        }
        wave_sync();
        sample_fine_wave(strip() + kStripAll, strip() + kStripW, a.NC, a.NF);   // strip[0, NT) = merged depths, ascending
        load_bias(a.net_f);

        // ---------------------------------------------------------------- stylised fine passes + compositing
        put_acc(RayAccum{1.0, 0.f, 0.f, 0.f, 0.f});
        for (int tile = 0; tile < tiles_f; ++tile) {
            const float* s_all = strip() + kStripAll;
            const int i = 16 * tile + (fresh_lane() & 15);
            float tt = s_all[i];
            const bool last = tile + 1 >= tiles_f;
            float col[3];
            // (the coarse weights are dead by now: their region of the strip takes one sigma per lane)
            styled_pass<CF>(smem, wave, fresh_lane(), sa, last ? a.net_c + kNerfBiasBytes : sa.concat_stream, o, d, tt, strip() + kStripZ,
                            lane_slab(), strip() + kStripW + fresh_lane(), col);
            const float sig = strip()[kStripW + fresh_lane()];
            const float* s_all2 = strip() + kStripAll;
            const int i2 = 16 * tile + (fresh_lane() & 15);
            tt = s_all2[i2];
            const float tn = s_all2[min(i2 + 1, NT - 1)];
            RayAccum acc = get_acc();
            composite_tile<true>(sig, col[0], col[1], col[2], tt, (i2 + 1 < NT) ? tn - tt : 1e10f, acc);
            put_acc(acc);
        }
        const RayAccum acc = get_acc();
        if (valid && fresh_lane() == 0) {
            a.rgb[ray * 3 + 0] = acc.r, a.rgb[ray * 3 + 1] = acc.g, a.rgb[ray * 3 + 2] = acc.b;
            a.t[ray] = acc.t;
        }
    }
    wait_vmcnt<0>();   // the look-ahead of the last pass must not outlive the workgroup's LDS
}

// n_wg: the number of slabs the style handle owns (one workgroup per slab)
int launch_fused_styled(int prec_c, const FusedStyledArgs& a, int n_wg, hipStream_t st) {
    const long long groups = (a.ray.R + 7) / 8;
    const unsigned grid = (unsigned)(groups < n_wg ? groups : n_wg);
    if (prec_c == TGTC_PREC_FP16X3)
        fused_styled_kernel<TGTC_PREC_FP16X3><<<grid, 512, 0, st>>>(a);
    else
        return fail(TGTC_ERR_UNSUPPORTED, "fused stylised render: no kernel for coarse precision %d", prec_c);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

// can the stylised ray kernel take this call?  (otherwise render.hip runs the per-sample chain)
bool fused_styled_supports(int prec_c, int prec_f, int prec_style, int n_coarse, int n_fine) {
    return prec_c == TGTC_PREC_FP16X3 && prec_f == TGTC_PREC_FP16X3 && prec_style == TGTC_PREC_FP16X3 &&
           fused_render_supports(prec_c, prec_f, n_coarse, n_fine);
}

}  // namespace tgtc
