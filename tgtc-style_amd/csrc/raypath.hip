// Per-ray kernels of the render hot path: ray generation (+NDC warp), coarse sampling, positional
// encoding, alpha compositing, inverse-CDF fine sampling, latent gather.
//
// These are HBM-bound elementwise / per-ray-scan kernels (a few bytes and a few dozen flops per
// element); the design rules that matter are coalesced access, one wavefront per ray for the scans,
// and LDS staging of the per-ray sample buffers.  The arithmetic follows the reference's evaluation
// order (no FMA contraction where the reference has separate roundings) so float64 outputs agree to
// the last bits and float32 outputs to ~1 ulp.
#include "common.h"
#include "raymarch_wave.h"

// The reference evaluates these formulas with separate multiplies and adds (numpy / ATen CPU);
// keep the same roundings.
#pragma clang fp contract(off)

namespace tgtc {

// ------------------------------------------------------------------------------------------- rays
// reference dataset.py:33-42 + dataset.py:44-61
struct RayGenArgs {
    int H, W;
    double fx, fy, cx, cy;
    float c2w[12];
    int pixel_alignment, ndc;
    double ndc_near;
    long long first, n;
};

__global__ void __launch_bounds__(256) gen_rays_kernel(RayGenArgs a, double* __restrict__ rays_o,
                                                       double* __restrict__ rays_d) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n) return;
    const long long pix = a.first + idx;
    float col = (float)(pix % a.W), row = (float)(pix / a.W);
    if (a.pixel_alignment) {
        col = col + 0.5f;
        row = row + 0.5f;
    }
    // dataset.py:37: float32 grid minus float64 intrinsics -> float64
    const double cam[3] = {((double)col - a.cx) / a.fx, (-((double)row - a.cy)) / a.fy, -1.0};
    double d[3], o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // dataset.py:39: sum over the last axis of dirs[...,None,:] * c2w[:3,:3]
        const double p0 = cam[0] * (double)a.c2w[4 * k + 0];
        const double p1 = cam[1] * (double)a.c2w[4 * k + 1];
        const double p2 = cam[2] * (double)a.c2w[4 * k + 2];
        d[k] = (p0 + p1) + p2;
        o[k] = (double)a.c2w[4 * k + 3];
    }
    if (a.ndc) {
        // dataset.py:46-47
        const double t = (-(a.ndc_near + o[2])) / d[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = o[k] + t * d[k];
        // dataset.py:50-56
        const double sx = -1.0 / ((double)a.W / (2.0 * a.fx));
        const double sy = -1.0 / ((double)a.H / (2.0 * a.fx));
        const double o0 = sx * o[0] / o[2];
        const double o1 = sy * o[1] / o[2];
        const double o2 = 1.0 + 2.0 * a.ndc_near / o[2];
        const double d0 = sx * (d[0] / d[2] - o[0] / o[2]);
        const double d1 = sy * (d[1] / d[2] - o[1] / o[2]);
        const double d2 = -2.0 * a.ndc_near / o[2];
        o[0] = o0, o[1] = o1, o[2] = o2;
        d[0] = d0, d[1] = d1, d[2] = d2;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_o[idx * 3 + k] = o[k];
        rays_d[idx * 3 + k] = d[k];
    }
}

// ------------------------------------------------------------------------------- coarse sampling
// reference utils.py:509-531
__global__ void __launch_bounds__(256) sample_coarse_kernel(const double* __restrict__ rays_o,
                                                            const double* __restrict__ rays_d, long long R, int N,
                                                            float near_, float far_, const float* __restrict__ jitter,
                                                            double* __restrict__ pts, float* __restrict__ ts) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * N) return;
    const long long r = idx / N;
    const int i = (int)(idx % N);
    const float t = jitter ? coarse_t_jittered(i, N, near_, far_, jitter[idx]) : coarse_t(i, N, near_, far_);
    ts[idx] = t;
    if (pts) {
#pragma unroll
        for (int k = 0; k < 3; ++k) pts[idx * 3 + k] = rays_o[r * 3 + k] + (double)t * rays_d[r * 3 + k];  // :529
    }
}

// ------------------------------------------------------------------------------------------ posenc
// reference models.py:46-60; output cast to float32 (models.py:219-220)
template <typename T>
__global__ void __launch_bounds__(256) posenc_kernel(const T* __restrict__ x, long long M, int L,
                                                     float* __restrict__ out) {
    const int width = 3 + 6 * L;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * width) return;
    const long long m = idx / width;
    const int c = (int)(idx % width);
    if (c < 3) {
        out[idx] = (float)x[m * 3 + c];
        return;
    }
    const int band = (c - 3) / 6, rem = (c - 3) % 6;
    const T arg = x[m * 3 + rem % 3] * (T)(double)(1 << band);
    out[idx] = (float)(rem < 3 ? sin(arg) : cos(arg));
}

// -------------------------------------------------------------------------------------- composite
// reference utils.py:354-386.  One wavefront per ray; lane l owns the contiguous run of samples
// [l*C, l*C+C), C = ceil(N/64).  The transmittance is the reference's SEQUENTIAL product, sample after sample in
// float64 and rounded to float32 per element (torch.cumprod on the CPU accumulates in acc_type<float> = double):
// every lane walks the whole chain with the `keep` factors broadcast lane by lane, so the weights do not depend
// on the tiling (raymarch_wave.h composite_tile computes the same chain 16 samples at a time).  Then four weighted
// wave reductions.
template <int C>
__device__ __forceinline__ void composite_wave(const float* __restrict__ rgb, const float* __restrict__ sigma,
                                               const float* __restrict__ ts, int N, float* rgb_exp, float* t_exp,
                                               float* weights, const float* __restrict__ noise = nullptr,
                                               int white_bkgd = 0) {
    const int lane = threadIdx.x & 63;
    float alpha[C], tv[C], keep[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int i = lane * C + k;
        alpha[k] = 0.0f, tv[k] = 0.0f, keep[k] = 1.0f;
        if (i < N) {
            const float t0 = ts[i];
            const float delta = (i + 1 < N) ? ts[i + 1] - t0 : 1e10f;       // utils.py:367-369
            const float raw = noise ? sigma[i] + noise[i] : sigma[i];      // training-time regulariser, :371-376
            const float dens = fmaxf(fmaxf(raw, 0.0f), 0.0f);               // relu(relu(.)) :365,:376
            alpha[k] = 1.0f - expf(-dens * delta);
            tv[k] = t0;
            keep[k] = 1.0f - alpha[k] + 1e-10f;                             // :378
        }
    }
    double run = 1.0;
    float trans[C];
    for (int l = 0; l < 64; ++l) {   // wave-uniform trip count; samples behind N carry keep = 1
#pragma unroll
        for (int k = 0; k < C; ++k) {
            if (lane == l) trans[k] = (float)run;
            run = run * (double)__shfl(keep[k], l);
        }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float wsum = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int i = lane * C + k;
        if (i < N) {
            const float w = alpha[k] * trans[k];
            wsum = wsum + w;
            if (weights) weights[i] = w;
            if (rgb) {  // wave-uniform: the sigma-only coarse pass of a fused render has no colours
                acc[0] = acc[0] + w * rgb[i * 3 + 0];
                acc[1] = acc[1] + w * rgb[i * 3 + 1];
                acc[2] = acc[2] + w * rgb[i * 3 + 2];
            }
            acc[3] = acc[3] + w * tv[k];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] + __shfl_xor(acc[j], off);
    }
    if (white_bkgd) {                                                       // rgb + (1 - acc_map), :381-384
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wsum = wsum + __shfl_xor(wsum, off);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = acc[j] + (1.0f - wsum);
    }
    if (lane == 0) {
        if (rgb_exp) rgb_exp[0] = acc[0], rgb_exp[1] = acc[1], rgb_exp[2] = acc[2];
        if (t_exp) t_exp[0] = acc[3];
    }
}

// Backward of alpha_composition (the training loops of the reference differentiate through it, train_tgtcs.py:218-309):
// given dL/d rgb_exp [3], dL/d t_exp, dL/d weights [N] of one ray, produce dL/d rgb [N,3] and dL/d sigma [N].
//   w_i = a_i T_i,  T_i = prod_{j<i} k_j,  k_j = 1 - a_j + 1e-10,  a_i = 1 - exp(-s_i d_i),  s_i = relu(relu(sigma_i + noise_i))
//   G_i = g_rgb . c_i + g_t t_i + g_w_i                      (dL/dw_i)
//   dL/da_i = G_i T_i - (sum_{m>i} G_m w_m) / k_i            (every later T_m carries the factor k_i)
//   dL/dsigma_i = dL/da_i * d_i (1 - a_i) * [sigma_i + noise_i > 0]
// One wavefront per ray, lane l owns samples [l*C, l*C+C); the suffix sum is a reverse wave scan.
template <int C>
__device__ __forceinline__ void composite_backward_wave(const float* __restrict__ rgb, const float* __restrict__ sigma,
                                                        const float* __restrict__ ts, const float* __restrict__ noise,
                                                        int N, int white_bkgd, const float* __restrict__ g_rgb,
                                                        const float* __restrict__ g_t, const float* __restrict__ g_w,
                                                        float* __restrict__ d_rgb, float* __restrict__ d_sigma) {
    const int lane = threadIdx.x & 63;
    float alpha[C], tv[C], keep[C], delta[C], live[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int i = lane * C + k;
        alpha[k] = 0.0f, tv[k] = 0.0f, keep[k] = 1.0f, delta[k] = 0.0f, live[k] = 0.0f;
        if (i < N) {
            const float t0 = ts[i];
            delta[k] = (i + 1 < N) ? ts[i + 1] - t0 : 1e10f;
            const float raw = noise ? sigma[i] + noise[i] : sigma[i];
            live[k] = raw > 0.0f ? 1.0f : 0.0f;
            alpha[k] = 1.0f - expf(-fmaxf(raw, 0.0f) * delta[k]);
            tv[k] = t0;
            keep[k] = 1.0f - alpha[k] + 1e-10f;
        }
    }
    double run = 1.0;
    float trans[C];
    for (int l = 0; l < 64; ++l) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            if (lane == l) trans[k] = (float)run;
            run = run * (double)__shfl(keep[k], l);
        }
    }
    const float gr = g_rgb ? g_rgb[0] : 0.0f, gg = g_rgb ? g_rgb[1] : 0.0f, gb = g_rgb ? g_rgb[2] : 0.0f;
    const float gt = g_t ? g_t[0] : 0.0f;
    // with a white background rgb_exp also carries -(sum w) in every channel: dL/dw_i gets -(gr + gg + gb)
    const float gwhite = white_bkgd ? -(gr + gg + gb) : 0.0f;
    float G[C], Gw[C], part = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int i = lane * C + k;
        G[k] = Gw[k] = 0.0f;
        if (i < N) {
            const float w = alpha[k] * trans[k];
            G[k] = gr * rgb[i * 3 + 0] + gg * rgb[i * 3 + 1] + gb * rgb[i * 3 + 2] + gt * tv[k] + (g_w ? g_w[i] : 0.0f) + gwhite;
            Gw[k] = G[k] * w;
            part += Gw[k];
            if (d_rgb) d_rgb[i * 3 + 0] = w * gr, d_rgb[i * 3 + 1] = w * gg, d_rgb[i * 3 + 2] = w * gb;
        }
    }
    // suffix sums: total of the lanes behind this one, then inside the run
    float incl = part;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float dn = __shfl_down(incl, off);
        if (lane + off < 64) incl = incl + dn;
    }
    float suffix = incl - part;   // sum over lanes > this lane
#pragma unroll
    for (int k = C - 1; k >= 0; --k) {
        const int i = lane * C + k;
        if (i < N && d_sigma) {
            const float d_alpha = G[k] * trans[k] - suffix / keep[k];
            d_sigma[i] = d_alpha * delta[k] * (1.0f - alpha[k]) * live[k];
        }
        suffix += Gw[k];
    }
}

template <int C>
__global__ void __launch_bounds__(256) composite_backward_kernel(const float* __restrict__ rgb, const float* __restrict__ sigma,
                                                                 const float* __restrict__ ts, const float* __restrict__ noise,
                                                                 long long R, int N, int white_bkgd,
                                                                 const float* __restrict__ g_rgb, const float* __restrict__ g_t,
                                                                 const float* __restrict__ g_w, float* __restrict__ d_rgb,
                                                                 float* __restrict__ d_sigma) {
    const long long r = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (r >= R) return;  // wave-uniform
    composite_backward_wave<C>(rgb + r * N * 3, sigma + r * N, ts + r * N, noise ? noise + r * N : nullptr, N, white_bkgd,
                               g_rgb ? g_rgb + r * 3 : nullptr, g_t ? g_t + r : nullptr, g_w ? g_w + r * N : nullptr,
                               d_rgb ? d_rgb + r * N * 3 : nullptr, d_sigma ? d_sigma + r * N : nullptr);
}

template <int C>
__global__ void __launch_bounds__(256) composite_kernel(const float* __restrict__ rgb, const float* __restrict__ sigma,
                                                        const float* __restrict__ ts, long long R, int N,
                                                        float* __restrict__ rgb_exp, float* __restrict__ t_exp,
                                                        float* __restrict__ weights, const float* __restrict__ noise,
                                                        int white_bkgd) {
    const long long r = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (r >= R) return;  // wave-uniform
    composite_wave<C>(rgb ? rgb + r * N * 3 : nullptr, sigma + r * N, ts + r * N, N,
                      rgb_exp ? rgb_exp + r * 3 : nullptr, t_exp ? t_exp + r : nullptr,
                      weights ? weights + r * N : nullptr, noise ? noise + r * N : nullptr, white_bkgd);
}

// ------------------------------------------------------------------------------------ fine sampling
// reference utils.py:573-609.  One wavefront per ray, per-ray buffers staged in LDS:
//   bins (N-1 midpoints), cdf (N-1), merged depths (N+n_fine).
// The cdf is a float64 running sum rounded to float32 per element, as ATen's CPU cumsum does
// (acc_type<float> = double).  The final torch.sort is realised as a stable rank (counting) sort,
// exact for any input order.
constexpr int kFineMaxN = 256;     // coarse samples per ray supported
constexpr int kFineMaxTotal = 512; // coarse + fine

__global__ void __launch_bounds__(64) sample_fine_kernel(const double* __restrict__ rays_o,
                                                         const double* __restrict__ rays_d,
                                                         const float* __restrict__ ts_in,
                                                         const float* __restrict__ w_in, long long R, int N,
                                                         int NF, double* __restrict__ pts_out,
                                                         float* __restrict__ ts_out) {
    __shared__ float s_bins[kFineMaxN];
    __shared__ float s_cdf[kFineMaxN];
    __shared__ float s_all[kFineMaxTotal];
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    const float* ts = ts_in + r * N;
    const float* w = w_in + r * N;
    const int B = N - 1;  // bins / cdf entries
    const int P = N - 2;  // pdf entries

    for (int i = lane; i < N; i += 64) s_all[i] = ts[i];
    for (int i = lane; i < B; i += 64) s_bins[i] = 0.5f * (ts[i + 1] + ts[i]);  // utils.py:574

    // sum of (w + 1e-5) over the interior weights (utils.py:575,584-585)
    double part = 0.0;
    for (int i = lane; i < P; i += 64) part += (double)(w[i + 1] + 1e-5f);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    const float total = (float)part;

    // cdf[0] = 0, cdf[i+1] = float(sum_{j<=i} pdf[j]) with a float64 running sum (utils.py:586-587).
    // Blocked scan: lane owns pdf entries [lane*C, lane*C+C).
    const int C = (P + 63) / 64;
    double run = 0.0;
    for (int k = 0; k < C; ++k) {
        const int i = lane * C + k;
        if (i < P) run += (double)((w[i + 1] + 1e-5f) / total);
    }
    double incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    double prefix = incl - run;
    if (lane == 0) s_cdf[0] = 0.0f;
    for (int k = 0; k < C; ++k) {
        const int i = lane * C + k;
        if (i < P) {
            prefix += (double)((w[i + 1] + 1e-5f) / total);
            s_cdf[i + 1] = (float)prefix;
        }
    }
    __syncthreads();

    // inverse CDF at u = linspace(0,1,NF) (utils.py:589-607)
    for (int j = lane; j < NF; j += 64) {
        const float u = linspace01(j, NF);
        int lo = 0, hi = B;  // searchsorted(right=True): first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, B - 1);
        const float c0 = s_cdf[below], c1 = s_cdf[above];
        const float b0 = s_bins[below], b1 = s_bins[above];
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.0f;
        const float t = (u - c0) / den;
        s_all[N + j] = b0 + t * (b1 - b0);
    }
    __syncthreads();

    // stable rank sort of the N+NF depths (utils.py:577)
    const int T = N + NF;
    const double o0 = rays_o[r * 3 + 0], o1 = rays_o[r * 3 + 1], o2 = rays_o[r * 3 + 2];
    const double d0 = rays_d[r * 3 + 0], d1 = rays_d[r * 3 + 1], d2 = rays_d[r * 3 + 2];
    for (int e = lane; e < T; e += 64) {
        const float v = s_all[e];
        int rank = 0;
        for (int q = 0; q < T; ++q) {
            const float x = s_all[q];
            rank += (x < v) || (x == v && q < e);
        }
        ts_out[r * T + rank] = v;
        if (pts_out) {
            double* p = pts_out + (r * T + rank) * 3;
            p[0] = o0 + d0 * (double)v;  // utils.py:578
            p[1] = o1 + d1 * (double)v;
            p[2] = o2 + d2 * (double)v;
        }
    }
}

// ------------------------------------------------------------------------------------------ latents
// reference models.py:490-506
__global__ void __launch_bounds__(256) latents_kernel(const float* __restrict__ latents, const float* __restrict__ mu,
                                                      int S, int F, int D, const long long* __restrict__ style_ids,
                                                      const long long* __restrict__ frame_ids, long long R,
                                                      float sigma_scale, int tile7, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * D) return;
    const long long r = idx / D;
    const int c = (int)(idx % D);
    const long long sid = style_ids[r];
    long long flat = sid * F + frame_ids[r];   // :492
    const long long rows = (long long)S * F;
    if (tile7) flat = flat % rows;             // table.repeat(7,1)[flat] for flat < 7*rows (:496)
    const float z = latents[flat * D + c];
    const float m = mu[sid * D + c];
    out[idx] = m + sigma_scale * (z - m);      // :504-505
}

// gradient of the same: d_latents[flat] += sigma_scale * g, d_mu[sid] += (1 - sigma_scale) * g (both pre-zeroed, atomics:
// many rays share a row)
__global__ void __launch_bounds__(256) latents_backward_kernel(const float* __restrict__ g, int S, int F, int D,
                                                               const long long* __restrict__ style_ids,
                                                               const long long* __restrict__ frame_ids, long long R,
                                                               float sigma_scale, int tile7, float* __restrict__ d_latents,
                                                               float* __restrict__ d_mu) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * D) return;
    const long long r = idx / D;
    const int c = (int)(idx % D);
    const long long sid = style_ids[r];
    long long flat = sid * F + frame_ids[r];
    if (tile7) flat = flat % ((long long)S * F);
    const float v = g[idx];
    if (d_latents) atomicAdd(d_latents + flat * D + c, sigma_scale * v);
    if (d_mu) atomicAdd(d_mu + sid * D + c, (1.0f - sigma_scale) * v);
}

}  // namespace tgtc

using namespace tgtc;

static inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" int tgtc_gen_rays(int H, int W, double fx, double fy, double cx, double cy, const float* c2w,
                             int pixel_alignment, int ndc, double ndc_near, int64_t first_pixel, int64_t n,
                             double* rays_o, double* rays_d, void* stream) {
    TGTC_REQUIRE(H > 0 && W > 0 && c2w, "gen_rays: bad argument");
    TGTC_REQUIRE(first_pixel >= 0 && n >= 0 && first_pixel + n <= (int64_t)H * W, "gen_rays: pixel range outside the frame");
    if (n == 0) return TGTC_OK;
    TGTC_REQUIRE(rays_o && rays_d, "gen_rays: null output");
    RayGenArgs a;
    a.H = H, a.W = W, a.fx = fx, a.fy = fy, a.cx = cx, a.cy = cy;
    for (int i = 0; i < 12; ++i) a.c2w[i] = c2w[i];
    a.pixel_alignment = pixel_alignment, a.ndc = ndc, a.ndc_near = ndc_near, a.first = first_pixel, a.n = n;
    gen_rays_kernel<<<blocks_for(n, 256), 256, 0, as_stream(stream)>>>(a, rays_o, rays_d);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_sample_coarse(const double* rays_o, const double* rays_d, int64_t R, int N, float near_,
                                  float far_, const float* jitter, double* pts, float* ts, void* stream) {
    TGTC_REQUIRE(R >= 0 && N >= 2, "sample_coarse: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(ts && (!pts || (rays_o && rays_d)), "sample_coarse: null pointer");
    sample_coarse_kernel<<<blocks_for(R * N, 256), 256, 0, as_stream(stream)>>>(rays_o, rays_d, R, N, near_, far_,
                                                                               jitter, pts, ts);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_posenc(const void* x, int x_is_f64, int64_t M, int L, float* out, void* stream) {
    TGTC_REQUIRE(M >= 0 && L >= 0 && L <= 30, "posenc: bad argument");
    if (M == 0) return TGTC_OK;
    TGTC_REQUIRE(x && out, "posenc: null pointer");
    const unsigned nb = blocks_for(M * (3 + 6 * L), 256);
    if (x_is_f64)
        posenc_kernel<double><<<nb, 256, 0, as_stream(stream)>>>((const double*)x, M, L, out);
    else
        posenc_kernel<float><<<nb, 256, 0, as_stream(stream)>>>((const float*)x, M, L, out);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

namespace tgtc {
int launch_composite_ex(const float* rgb, const float* sigma, const float* ts, int64_t R, int N, float* rgb_exp,
                        float* t_exp, float* weights, hipStream_t st, const float* noise, int white_bkgd) {
    const unsigned nb = blocks_for(R, 4);
    const int C = (N + 63) / 64;
#define TGTC_COMPOSITE(CC) composite_kernel<CC><<<nb, 256, 0, st>>>(rgb, sigma, ts, R, N, rgb_exp, t_exp, weights, noise, white_bkgd)
    switch (C) {
        case 1: TGTC_COMPOSITE(1); break;
        case 2: TGTC_COMPOSITE(2); break;
        case 3: TGTC_COMPOSITE(3); break;
        case 4: TGTC_COMPOSITE(4); break;
        case 5: case 6: case 7: case 8: TGTC_COMPOSITE(8); break;
        default: return fail(TGTC_ERR_UNSUPPORTED, "composite: N=%d samples per ray exceeds 512", N);
    }
#undef TGTC_COMPOSITE
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

int launch_composite(const float* rgb, const float* sigma, const float* ts, int64_t R, int N, float* rgb_exp,
                     float* t_exp, float* weights, hipStream_t st) {
    return launch_composite_ex(rgb, sigma, ts, R, N, rgb_exp, t_exp, weights, st, nullptr, 0);
}

int launch_composite_backward(const float* rgb, const float* sigma, const float* ts, const float* noise, int64_t R, int N,
                              int white_bkgd, const float* g_rgb, const float* g_t, const float* g_w, float* d_rgb,
                              float* d_sigma, hipStream_t st) {
    const unsigned nb = blocks_for(R, 4);
    const int C = (N + 63) / 64;
#define TGTC_COMPOSITE_BWD(CC) \
    composite_backward_kernel<CC><<<nb, 256, 0, st>>>(rgb, sigma, ts, noise, R, N, white_bkgd, g_rgb, g_t, g_w, d_rgb, d_sigma)
    switch (C) {
        case 1: TGTC_COMPOSITE_BWD(1); break;
        case 2: TGTC_COMPOSITE_BWD(2); break;
        case 3: TGTC_COMPOSITE_BWD(3); break;
        case 4: TGTC_COMPOSITE_BWD(4); break;
        case 5: case 6: case 7: case 8: TGTC_COMPOSITE_BWD(8); break;
        default: return fail(TGTC_ERR_UNSUPPORTED, "composite_backward: N=%d samples per ray exceeds 512", N);
    }
#undef TGTC_COMPOSITE_BWD
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

int launch_sample_fine(const double* rays_o, const double* rays_d, const float* ts, const float* weights, int64_t R,
                       int N, int n_fine, double* pts_out, float* ts_out, hipStream_t st) {
    if (N < 3 || N > kFineMaxN || n_fine < 1 || N + n_fine > kFineMaxTotal)
        return fail(TGTC_ERR_UNSUPPORTED, "sample_fine: N=%d n_fine=%d outside [3,%d] / total<=%d", N, n_fine,
                    kFineMaxN, kFineMaxTotal);
    sample_fine_kernel<<<(unsigned)R, 64, 0, st>>>(rays_o, rays_d, ts, weights, R, N, n_fine, pts_out, ts_out);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}
}  // namespace tgtc

extern "C" int tgtc_composite(const float* rgb, const float* sigma, const float* ts, int64_t R, int N,
                              float* rgb_exp, float* t_exp, float* weights, void* stream) {
    TGTC_REQUIRE(R >= 0 && N >= 1, "composite: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rgb && sigma && ts && rgb_exp && t_exp, "composite: null pointer");
    return launch_composite(rgb, sigma, ts, R, N, rgb_exp, t_exp, weights, as_stream(stream));
}

extern "C" int tgtc_composite_train(const float* rgb, const float* sigma, const float* ts, const float* noise,
                                    int white_bkgd, int64_t R, int N, float* rgb_exp, float* t_exp, float* weights,
                                    void* stream) {
    TGTC_REQUIRE(R >= 0 && N >= 1, "composite_train: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rgb && sigma && ts && rgb_exp && t_exp, "composite_train: null pointer");
    return launch_composite_ex(rgb, sigma, ts, R, N, rgb_exp, t_exp, weights, as_stream(stream), noise, white_bkgd);
}

extern "C" int tgtc_composite_backward(const float* rgb, const float* sigma, const float* ts, const float* noise,
                                       int white_bkgd, int64_t R, int N, const float* grad_rgb_exp,
                                       const float* grad_t_exp, const float* grad_weights, float* grad_rgb,
                                       float* grad_sigma, void* stream) {
    TGTC_REQUIRE(R >= 0 && N >= 1, "composite_backward: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rgb && sigma && ts && (grad_rgb || grad_sigma), "composite_backward: null pointer");
    return launch_composite_backward(rgb, sigma, ts, noise, R, N, white_bkgd, grad_rgb_exp, grad_t_exp, grad_weights, grad_rgb,
                                     grad_sigma, as_stream(stream));
}

extern "C" int tgtc_sample_fine(const double* rays_o, const double* rays_d, const float* ts, const float* weights,
                                int64_t R, int N, int n_fine, double* pts_out, float* ts_out, void* stream) {
    TGTC_REQUIRE(R >= 0, "sample_fine: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rays_o && rays_d && ts && weights && ts_out, "sample_fine: null pointer");
    return launch_sample_fine(rays_o, rays_d, ts, weights, R, N, n_fine, pts_out, ts_out, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------ image epilogue
// One workgroup per frame finds min / max of the depth (exact: min/max are order independent), then the same
// workgroup converts its frame.  numpy semantics: float32 arithmetic throughout (x*255, the division), float ->
// int32 truncates toward zero, int32 -> uint8 keeps the low byte.
__global__ void __launch_bounds__(1024) image_epilogue_kernel(const float* __restrict__ rgb, const float* __restrict__ t,
                                                             long long pixels, float eps, unsigned char* __restrict__ rgb8,
                                                             unsigned char* __restrict__ depth8) {
    __shared__ float s_min[16], s_max[16];
    const long long base = (long long)blockIdx.x * pixels;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (depth8) {
        float lo = __builtin_inff(), hi = -__builtin_inff();
        for (long long i = tid; i < pixels; i += 1024) {
            const float v = t[base + i];
            lo = fminf(lo, v), hi = fmaxf(hi, v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lo = fminf(lo, __shfl_xor(lo, o)), hi = fmaxf(hi, __shfl_xor(hi, o));
        if (lane == 0) s_min[wave] = lo, s_max[wave] = hi;
        __syncthreads();
        lo = s_min[0], hi = s_max[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) lo = fminf(lo, s_min[w]), hi = fmaxf(hi, s_max[w]);
        const float den = __fadd_rn(__fsub_rn(hi, lo), eps);
        for (long long i = tid; i < pixels; i += 1024) {
            const float v = __fdiv_rn(__fsub_rn(t[base + i], lo), den);   // IEEE division, as numpy
            depth8[base + i] = (unsigned char)(int)__fmul_rn(v, 255.0f);
        }
    }
    if (rgb8) {
        for (long long i = tid; i < pixels * 3; i += 1024) rgb8[base * 3 + i] = (unsigned char)(int)__fmul_rn(rgb[base * 3 + i], 255.0f);
    }
}

extern "C" int tgtc_image_epilogue(const float* rgb, const float* t, int64_t frames, int64_t pixels, float eps,
                                   unsigned char* rgb8, unsigned char* depth8, void* stream) {
    TGTC_REQUIRE(frames >= 0 && pixels >= 0 && frames < (1LL << 31), "image_epilogue: bad argument");
    if (frames == 0 || pixels == 0) return TGTC_OK;
    TGTC_REQUIRE((!rgb8 || rgb) && (!depth8 || t) && (rgb8 || depth8), "image_epilogue: null pointer");
    image_epilogue_kernel<<<(unsigned)frames, 1024, 0, as_stream(stream)>>>(rgb, t, pixels, eps, rgb8, depth8);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_latents_backward(const float* grad_out, int S, int F, int D, const int64_t* style_ids,
                                     const int64_t* frame_ids, int64_t R, float sigma_scale, int tile7, float* d_latents,
                                     float* d_mu, void* stream) {
    TGTC_REQUIRE(S > 0 && F > 0 && D > 0 && R >= 0, "latents_backward: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(grad_out && style_ids && frame_ids, "latents_backward: null pointer");
    latents_backward_kernel<<<blocks_for(R * D, 256), 256, 0, as_stream(stream)>>>(
        grad_out, S, F, D, (const long long*)style_ids, (const long long*)frame_ids, R, sigma_scale, tile7, d_latents, d_mu);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_latents_forward(const float* latents, const float* mu, int S, int F, int D,
                                    const int64_t* style_ids, const int64_t* frame_ids, int64_t R, float sigma_scale,
                                    int tile7, float* out, void* stream) {
    TGTC_REQUIRE(S > 0 && F > 0 && D > 0 && R >= 0, "latents_forward: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(latents && mu && style_ids && frame_ids && out, "latents_forward: null pointer");
    latents_kernel<<<blocks_for(R * D, 256), 256, 0, as_stream(stream)>>>(
        latents, mu, S, F, D, (const long long*)style_ids, (const long long*)frame_ids, R, sigma_scale, tile7, out);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}
