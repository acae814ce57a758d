// Per-ray arithmetic of the render hot path as device functions, shared by the stand-alone per-ray kernels
// (raypath.hip) and the fused ray kernel (render_fused.hip):
//   coarse depths        reference utils.py:509-531 (sampling_pts_uniform)
//   alpha compositing    reference utils.py:354-386 (alpha_composition), one 16-sample tile at a time
//   fine sampling        reference utils.py:573-609 (sampling_pts_fine_torch / sample_pdf, det=True)
// The reference evaluates these formulas with separate multiplies and adds (numpy / ATen CPU): FMA contraction is
// switched off for this header, whatever the including translation unit uses.
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_core.h"

#pragma clang fp contract(off)

namespace tgtc {

__device__ __forceinline__ float linspace01(int i, int n) {
    // torch.linspace(0,1,n) float32 (ATen RangeFactoriesKernel: first half start+step*i, second half
    // end-step*(n-1-i), each with a single rounding).
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? step * (float)i : __fmaf_rn(-step, (float)(n - 1 - i), 1.0f);
}

__device__ __forceinline__ float coarse_t(int i, int n, float near_, float far_) {
    float t = linspace01(i, n);
    return t * (far_ - near_) + near_;  // utils.py:514
}

__device__ __forceinline__ float coarse_t_jittered(int i, int n, float near_, float far_, float u) {
    // utils.py:521-524: interval between the midpoints to the neighbours
    const float t = coarse_t(i, n, near_, far_);
    const float lo = (i == 0) ? t : (t + coarse_t(i - 1, n, near_, far_)) / 2.0f;
    const float hi = (i == n - 1) ? t : (coarse_t(i + 1, n, near_, far_) + t) / 2.0f;
    return lo + (hi - lo) * u;
}

// Lanes of one wavefront exchange data through LDS without a workgroup barrier: the LDS executes a wave's
// instructions in order, so only the COMPILER has to be kept from moving accesses across the hand-over.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Running state of one ray's compositing, identical in all lanes of the wave.
struct RayAccum {
    double trans;      // prod_{j<i} (1 - alpha_j + 1e-10), utils.py:378 -- a float64 running product rounded to float32
                       // per element, which is what torch.cumprod computes on the CPU (ATen accumulates in acc_type)
    float r, g, b, t;  // sum w*c, sum w*t (utils.py:380-382)
};

// One tile of 16 consecutive samples of a ray, sample i = i0 + (lane & 15): lanes 0..15 carry sigma (and rgb).
//   delta = t_next - t, or 1e10 behind the ray's last sample (utils.py:367-369)
// Returns the sample's weight alpha * T (utils.py:379); updates acc with this tile's 16 samples.  Every lane of the
// wave calls it (row-wide shuffles); lanes 16..63 compute on copies of zero rows and their results are unused.
// The transmittance is the SEQUENTIAL product of the reference (sample after sample, in float64), so the weights --
// and with them the inverse-CDF samples, which are discontinuous in the weights -- do not depend on how a kernel tiles a ray.
// Lane K of this lane's row of 16 / the row rotated by R lanes, as DPP moves (row_newbcast, row_ror): no LDS, no address
// registers.  (`__shfl(v, k, 16)` builds sixteen per-lane ds_bpermute addresses from the lane id, which hipcc hoists out of a
// caller's tile loop and keeps -- or spills -- across the passes.)  Rotations by 8, 4, 2, 1 add up a row exactly like the
// xor butterfly: the partial sums of step s have period 16 >> s, so lane i + r and lane i ^ r hold the same value.
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, false));
}
template <int R>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + R, 0xf, 0xf, false));
}

template <bool COLOUR>
__device__ __forceinline__ float composite_tile(float sigma, float cr, float cg, float cb, float t, float delta, RayAccum& acc) {
    const int n = fresh_lane_id() & 15;   // read in place: not hoistable out of a caller's loop (render_fused.hip)
    const float dens = fmaxf(fmaxf(sigma, 0.0f), 0.0f);          // relu(relu(.)) utils.py:365,376
    const float alpha = 1.0f - expf(-dens * delta);
    const float keep = 1.0f - alpha + 1e-10f;
    double run = acc.trans, mine = acc.trans;
    static_for<16>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        if (n == k) mine = run;
        run = run * (double)row_bcast<k>(keep);
    });
    acc.trans = run;
    const float w = alpha * (float)mine;
    float s0 = COLOUR ? w * cr : 0.0f, s1 = COLOUR ? w * cg : 0.0f, s2 = COLOUR ? w * cb : 0.0f, s3 = w * t;
    static_for<4>([&](auto s_) {
        constexpr int off = 8 >> decltype(s_)::value;
        if constexpr (COLOUR) s0 = s0 + row_ror<off>(s0), s1 = s1 + row_ror<off>(s1), s2 = s2 + row_ror<off>(s2);
        s3 = s3 + row_ror<off>(s3);
    });
    acc.r = acc.r + s0, acc.g = acc.g + s1, acc.b = acc.b + s2, acc.t = acc.t + s3;
    return w;
}

// Fine sampling of ONE ray by ONE wavefront (wave-synchronous, no workgroup barrier):
//   s_all[0..N)  the ray's coarse depths (ascending)          -> on return s_all[0..N+NF) = sort(cat(ts, new samples))
//   s_w[0..N)    the coarse weights; reused for the cdf (N-1 entries)
// The cdf is a float64 running sum rounded to float32 per element, as ATen's CPU cumsum does (acc_type<float> =
// double); the final torch.sort is realised as a stable rank (counting) sort, exact for any input order.
// N <= 256, N + NF <= 256.
__device__ __forceinline__ void sample_fine_wave(float* s_all, float* s_w, int N, int NF) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));   // not hoistable out of a caller's loop (render_fused.hip)
    const int B = N - 1;  // bins / cdf entries
    const int P = N - 2;  // pdf entries
    const int C = (P + 63) / 64;

    // sum of (w + 1e-5) over the interior weights (utils.py:575,584-585)
    double part = 0.0;
    for (int i = lane; i < P; i += 64) part += (double)(s_w[i + 1] + 1e-5f);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    const float total = (float)part;

    // cdf[0] = 0, cdf[i+1] = float(sum_{j<=i} pdf[j]) with a float64 running sum (utils.py:586-587).
    // Blocked scan: lane owns pdf entries [lane*C, lane*C+C); they are read into registers before any lane
    // overwrites the weights with the cdf.
    float pdf[4];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane * C + k;
        pdf[k] = 0.0f;
        if (k < C && i < P) {
            pdf[k] = (s_w[i + 1] + 1e-5f) / total;
            run += (double)pdf[k];
        }
    }
    double incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    double prefix = incl - run;
    wave_sync();
    if (lane == 0) s_w[0] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane * C + k;
        if (k < C && i < P) {
            prefix += (double)pdf[k];
            s_w[i + 1] = (float)prefix;
        }
    }
    wave_sync();

    // inverse CDF at u = linspace(0,1,NF) (utils.py:589-607); bins = midpoints of the coarse depths (utils.py:574)
    for (int j = lane; j < NF; j += 64) {
        const float u = linspace01(j, NF);
        int lo = 0, hi = B;  // searchsorted(right=True): first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_w[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, B - 1);
        const float c0 = s_w[below], c1 = s_w[above];
        const float b0 = 0.5f * (s_all[below + 1] + s_all[below]), b1 = 0.5f * (s_all[above + 1] + s_all[above]);
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.0f;
        const float tt = (u - c0) / den;
        s_all[N + j] = b0 + tt * (b1 - b0);
    }
    wave_sync();

    // stable rank sort of the N+NF depths (utils.py:577), in place: every lane ranks its elements first
    const int T = N + NF;
    float v[4];
    int rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = lane + 64 * k;
        v[k] = 0.0f, rank[k] = 0;
        if (e < T) {
            v[k] = s_all[e];
            for (int q = 0; q < T; ++q) {
                const float x = s_all[q];
                rank[k] += (x < v[k]) || (x == v[k] && q < e);
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (lane + 64 * k < T) s_all[rank[k]] = v[k];
    wave_sync();
}

}  // namespace tgtc

#pragma clang fp contract(on)
