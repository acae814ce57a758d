// The whole plain render of a ray in ONE persistent kernel (reference rendering.py:27-51 cal_geometry; utils.py:509-531,
// 354-386, 573-609; models.py:46-60,95-117,216-223):
//
//   coarse depths -> [PE + coarse NeRF MLP (sigma)] x Nc/16 tiles -> weights -> inverse-CDF fine sampling + merge
//   -> [PE + fine NeRF MLP (rgb, sigma)] x (Nc+Nf)/16 tiles -> alpha compositing -> rgb, depth
//
// A wavefront owns a RAY; its column tiles are consecutive 16-sample tiles of that ray, so the transmittance and
// the weighted colour / depth sums of alpha compositing stay in registers across the passes over a ray's tiles,
// the per-ray sample buffers (depths, weights / cdf) live in a 1.25 KiB LDS strip per wave, and HBM sees 48 bytes
// of ray in and 16 bytes of pixel out -- no per-sample tensor exists anywhere.  The eight waves of a workgroup
// walk the weight streams in lockstep through one LDS ring that never drains (mlp_core.h, WeightStream PERSIST):
// coarse stream x Nc/16/NCT passes, fine stream x (Nc+Nf)/16/NCT passes, next group of eight rays.  The grid is
// one workgroup per CU; ray groups are dealt round-robin.
//
// Per-tile arithmetic is the code of the per-sample kernels (mlp_nerf_chain.h, mlp_nerf_mx_chain.h,
// raymarch_wave.h), so the result of a ray does not depend on which wave, workgroup or launch renders it.
#include "render_fused.h"

namespace tgtc {

template <int PC, int PF>
__global__ void __launch_bounds__(512, 2) fused_render_kernel(FusedArgs a) {
    constexpr bool DEPTHS = PF == kFusedDepthsOnly;   // coarse passes + fine sampling only: the stylised fine pass follows in its own kernel
    using CC = typename FusedCfg<PC>::C;
    using CF = typename FusedCfg<(DEPTHS ? PC : PF)>::C;
    static_assert(CC::NWAVES == 8 && CF::NWAVES == 8 && CC::SLOTS == CF::SLOTS, "one ring, eight waves");
    constexpr int NW = 8;

    // ring | bias table of the running phase (+ row exponents) | per-wave strips
    __shared__ __attribute__((aligned(16))) char smem[CC::RING_BYTES + kNerfBiasBytes + NW * kFusedStripBytes];

    // The lane id is re-read from the hardware (fresh_lane_id, mlp_core.h) wherever per-lane addresses are formed: a kernel-long
    // `threadIdx.x & 63` -- and whatever the optimiser hoists out of the ray loop -- stays live across the passes and is
    // spilled (see fused_pass), or evicted and restored around the asm blocks of the fp16mx pass (mx_asm_nerf.inc)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef TGTC_FUSED_PRIO
    // experiment (CDNA guide T5, static form): the younger half of an 8-wave workgroup loses VALU arbitration on every segment
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    auto fresh_lane = [&] { return fresh_lane_id(); };
    auto strip = [&]() -> float* {
        return reinterpret_cast<float*>(smem + CC::RING_BYTES + kNerfBiasBytes + wave * kFusedStripBytes);
    };
    // bias table (+ row exponents) of a phase: everyone is done with the old one, 2 KiB per wave, drained; the
    // barrier of the next enter() publishes it
    auto load_bias = [&](const char* net) {
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < kNerfBiasBytes / (NW * 1024); ++j)
            lds_dma16_t<(PC == TGTC_PREC_FP16_FP6 || PF == TGTC_PREC_FP16_FP6)>(
                net + (j * NW + wave) * 1024 + fresh_lane() * 16, smem + CC::RING_BYTES + (j * NW + wave) * 1024);
        wait_vmcnt<0>();
    };
    // the compositing state lives in the strip between passes (all lanes hold the same values)
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

    {   // chunks 0 .. SLOTS-2 of the first coarse pass: the state every enter() expects
        typename FusedStream<PC, false>::type first;
        const char* const one[1] = {a.net_c + kNerfBiasBytes};
        first.init(one, smem, wave, fresh_lane());
        if constexpr (PC == TGTC_PREC_FP16_FP6) {
            first.ring.next = first.ring.src[0];
            first.ring.persist_prologue();
        } else {
            first.next = first.src[0];
            first.persist_prologue();
        }
    }

    const int tiles_c = a.NC / 16, tiles_f = (a.NC + a.NF) / 16, NT = a.NC + a.NF;
    const long long groups = (a.R + NW - 1) / NW;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const long long ray_raw = grp * NW + wave;
        const bool valid = ray_raw < a.R;
        const long long ray = valid ? ray_raw : a.R - 1;   // tail: duplicate the last ray, the store is masked
        // the ray is wave-uniform: keep it in scalar registers for the 20 passes it lives through
        double o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = uniform_f64(a.rays_o[ray * 3 + k]), d[k] = uniform_f64(a.rays_d[ray * 3 + k]);
        if (a.jitter) {   // the jitter of sample i waits in s_w[i] until the sample's weight replaces it
            float* s_w = strip() + kStripW;
            for (int i = fresh_lane(); i < a.NC; i += 64) s_w[i] = a.jitter[ray * a.NC + i];
        }
        load_bias(a.net_c);

        // ---------------------------------------------------------------- coarse passes (sigma only)
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
            fused_pass<PC, false>(smem, wave, fresh_lane(), a.net_c, ((last && !DEPTHS) ? a.net_f : a.net_c) + kNerfBiasBytes, o, d, tt, sig, col);
            depths(tt, tn);   // recomputed rather than kept across the pass
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
        if constexpr (DEPTHS) {
            wave_sync();
            if (valid)
                for (int i = fresh_lane(); i < NT; i += 64) a.ts_out[ray * NT + i] = strip()[kStripAll + i];
            wave_sync();
        } else {
        load_bias(a.net_f);

        // ---------------------------------------------------------------- fine passes (rgb, sigma) + compositing
        put_acc(RayAccum{1.0, 0.f, 0.f, 0.f, 0.f});
        for (int tile = 0; tile < tiles_f; tile += CF::NCT) {
            auto depths = [&](float (&tt)[CF::NCT], float (&tn)[CF::NCT]) {
                const float* s_all = strip() + kStripAll;
                const int n = fresh_lane() & 15;
#pragma unroll
                for (int c = 0; c < CF::NCT; ++c) {
                    const int i = 16 * (tile + c) + n;
                    tt[c] = s_all[i], tn[c] = s_all[min(i + 1, NT - 1)];
                }
            };
            float tt[CF::NCT], tn[CF::NCT], sig[CF::NCT], col[CF::NCT][3];
            depths(tt, tn);
            const bool last = tile + CF::NCT >= tiles_f;
            fused_pass<PF, true>(smem, wave, fresh_lane(), a.net_f, (last ? a.net_c : a.net_f) + kNerfBiasBytes, o, d, tt, sig, col);
            depths(tt, tn);
            RayAccum acc = get_acc();
            const int n = fresh_lane() & 15;
#pragma unroll
            for (int c = 0; c < CF::NCT; ++c) {
                const int i = 16 * (tile + c) + n;
                composite_tile<true>(sig[c], col[c][0], col[c][1], col[c][2], tt[c], (i + 1 < NT) ? tn[c] - tt[c] : 1e10f, acc);
            }
            put_acc(acc);
        }
        const RayAccum acc = get_acc();
        if (valid && fresh_lane() == 0) {
            a.rgb[ray * 3 + 0] = acc.r, a.rgb[ray * 3 + 1] = acc.g, a.rgb[ray * 3 + 2] = acc.b;
            a.t[ray] = acc.t;
        }
        }   // !DEPTHS
    }
    wait_vmcnt<0>();   // the look-ahead of the last pass must not outlive the workgroup's LDS
}

int launch_fused_render(int prec_c, int prec_f, const FusedArgs& a, hipStream_t st) {
    int dev = 0, cus = 0;
    TGTC_HIP_CHECK(hipGetDevice(&dev));
    TGTC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long long groups = (a.R + 7) / 8;
    const unsigned grid = (unsigned)(groups < cus ? groups : cus);
    if (prec_c == TGTC_PREC_FP16X3 && prec_f == TGTC_PREC_FP16X3)
        fused_render_kernel<TGTC_PREC_FP16X3, TGTC_PREC_FP16X3><<<grid, 512, 0, st>>>(a);
    else if (prec_c == TGTC_PREC_FP16X3 && prec_f == TGTC_PREC_FP16_FP6)
        fused_render_kernel<TGTC_PREC_FP16X3, TGTC_PREC_FP16_FP6><<<grid, 512, 0, st>>>(a);
    else if (prec_c == TGTC_PREC_FP16 && prec_f == TGTC_PREC_FP16)
        fused_render_kernel<TGTC_PREC_FP16, TGTC_PREC_FP16><<<grid, 512, 0, st>>>(a);
    else
        return fail(TGTC_ERR_UNSUPPORTED, "fused render: no kernel for precisions %d (coarse) + %d (fine)", prec_c, prec_f);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

// Coarse passes + fine sampling of the fused kernel alone: ts_out[R, NC + NF] = the merged depths the fine pass is evaluated
// at (rendering.py:118-160 up to the second sample_pdf; the stylised fine pass then runs mlp_style.hip's kernel on them).
int launch_fused_depths(int prec_c, const FusedArgs& a, hipStream_t st) {
    int dev = 0, cus = 0;
    TGTC_HIP_CHECK(hipGetDevice(&dev));
    TGTC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long long groups = (a.R + 7) / 8;
    const unsigned grid = (unsigned)(groups < cus ? groups : cus);
    if (prec_c == TGTC_PREC_FP16X3)
        fused_render_kernel<TGTC_PREC_FP16X3, kFusedDepthsOnly><<<grid, 512, 0, st>>>(a);
    else if (prec_c == TGTC_PREC_FP16)
        fused_render_kernel<TGTC_PREC_FP16, kFusedDepthsOnly><<<grid, 512, 0, st>>>(a);
    else
        return fail(TGTC_ERR_UNSUPPORTED, "fused depths: no kernel for coarse precision %d", prec_c);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}
bool fused_depths_supports(int prec_c, int n_coarse, int n_fine) {
    return fused_render_supports(prec_c, prec_c, n_coarse, n_fine);
}

// can the fused kernel take this call?  (otherwise render.hip runs the chain of per-sample kernels)
bool fused_render_supports(int prec_c, int prec_f, int n_coarse, int n_fine) {
    const bool pair = (prec_c == TGTC_PREC_FP16X3 && (prec_f == TGTC_PREC_FP16X3 || prec_f == TGTC_PREC_FP16_FP6)) ||
                      (prec_c == TGTC_PREC_FP16 && prec_f == TGTC_PREC_FP16);
    const int step = prec_c == TGTC_PREC_FP16 ? 32 : 16;   // tiles per pass x 16 samples
    return pair && n_fine >= 1 && n_coarse >= 16 && n_coarse % step == 0 && (n_coarse + n_fine) % step == 0 && n_coarse <= 192 &&
           n_coarse + n_fine <= kFusedMaxTotal;
}

}  // namespace tgtc
