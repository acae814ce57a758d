// Argument block and input stage shared by the NeRF kernels (mlp_nerf.hip: fp16 / fp16x3, mlp_nerf_mx.hip:
// fp16+fp6): sample positions from rays / points, positional encodings straight into B fragments.
#pragma once
#include "mlp_core.h"

namespace tgtc {

enum InMode { IN_RAYS = 0, IN_PTS = 1, IN_ENC = 2 };

struct NerfArgs {
    const char* bias;    // device: padded bias table
    const char* stream;  // device: weight fragment stream
    long long M;         // samples
    int N;               // samples per ray (IN_RAYS)
    // inputs
    const double* rays_o;
    const double* rays_d;
    const float* ts;
    const double* pts;
    const double* dirs;
    const float* pts_enc;
    const float* dirs_enc;
    // outputs (any may be null)
    float* rgb;
    float* sigma;
    float* remap;
    float* out_pts_enc;
    float* out_dirs_enc;
};

// Ordinary global loads of a wave's NCT x 16 samples.  Must run BEFORE any LDS-DMA is issued: once a
// global_load_lds is in flight hipcc drains vmcnt(0) -- the whole prefetch -- at the first use of a loaded value.
template <int NCT, int IN_MODE>
__device__ __forceinline__ void nerf_load_samples(const NerfArgs& a, long long s_wave, int n, double (&pos)[NCT][3],
                                                  double (&dir)[NCT][3], long long (&sidx)[NCT]) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        long long s = s_wave + c * 16 + n;
        sidx[c] = s;
        if (s >= a.M) s = a.M - 1;  // tail: duplicate the last sample, stores are masked
        if constexpr (IN_MODE == IN_RAYS) {
            const long long r = (unsigned)s / (unsigned)a.N;  // M < 2^31 is checked at launch
            const double t = (double)a.ts[s];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dir[c][k] = a.rays_d[r * 3 + k];
                pos[c][k] = a.rays_o[r * 3 + k] + t * dir[c][k];  // rendering.py:27 / utils.py:529
            }
        } else if constexpr (IN_MODE == IN_PTS) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pos[c][k] = a.pts[s * 3 + k], dir[c][k] = a.dirs[s * 3 + k];
        }
        if constexpr (IN_MODE != IN_ENC) {
            // retire the loads here
#pragma unroll
            for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(pos[c][k]), "+v"(dir[c][k]));
        }
    }
}

// already-encoded inputs (MLP_style.forward boundary) -> B fragments; also ordinary loads, same rule
template <int NCT, bool SPLIT, bool FULL>
__device__ __forceinline__ void nerf_load_encoded(const NerfArgs& a, const long long (&sidx)[NCT], int g,
                                                  half8 (&pe_h)[2][NCT], half8 (&pe_l)[2][NCT], half8 (&de_h)[1][NCT],
                                                  half8 (&de_l)[1][NCT]) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const long long s = sidx[c] < a.M ? sidx[c] : a.M - 1;
        half8 h2[2], l2[2];
        load_encoded_point<SPLIT>(a.pts_enc + s * 63, g, h2, l2);
        pe_h[0][c] = h2[0], pe_h[1][c] = h2[1], pe_l[0][c] = l2[0], pe_l[1][c] = l2[1];
        if constexpr (FULL) load_encoded_dir<SPLIT>(a.dirs_enc + s * 27, g, de_h[0][c], de_l[0][c]);
    }
}

// positional encodings of positions / directions into B fragments (overlaps the weight prefetch latency)
template <int NCT, bool SPLIT, bool FULL>
__device__ __forceinline__ void nerf_encode(const NerfArgs& a, const double (&pos)[NCT][3], const double (&dir)[NCT][3],
                                            const long long (&sidx)[NCT], int g, half8 (&pe_h)[2][NCT],
                                            half8 (&pe_l)[2][NCT], half8 (&de_h)[1][NCT], half8 (&de_l)[1][NCT]) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const bool live = sidx[c] < a.M;
        half8 h2[2], l2[2];
        encode_point<SPLIT, SPLIT>(pos[c], g, h2, l2, (a.out_pts_enc && live) ? a.out_pts_enc + sidx[c] * 63 : nullptr);
        pe_h[0][c] = h2[0], pe_h[1][c] = h2[1], pe_l[0][c] = l2[0], pe_l[1][c] = l2[1];
        if constexpr (FULL) {
            // consecutive tiles of a wave are samples of the same ray almost always (rays of N = 16 k samples): the direction's
            // encoding is then the previous tile's, bit for bit, and a lone wave has nobody to hide four more sincos behind
            bool same = false;
            if constexpr (NCT > 1) {
                if (c > 0 && !a.out_dirs_enc)
                    same = __all(dir[c][0] == dir[c - 1][0] && dir[c][1] == dir[c - 1][1] && dir[c][2] == dir[c - 1][2]);
            }
            if (same) {
                de_h[0][c] = de_h[0][c > 0 ? c - 1 : 0], de_l[0][c] = de_l[0][c > 0 ? c - 1 : 0];
            } else {
                encode_dir<SPLIT, SPLIT>(dir[c], g, de_h[0][c], de_l[0][c],
                                         (a.out_dirs_enc && live) ? a.out_dirs_enc + sidx[c] * 27 : nullptr);
            }
        }
    }
}

// Late direction encoding (kernels short of registers: the 27-wide direction encoding is only needed by the colour
// head, eleven layers after the prologue): reload the direction of this lane's sample and encode it there.
template <int IN_MODE, bool SPLIT>
__device__ __forceinline__ void nerf_encode_dir_late(const NerfArgs& a, long long sidx, int g, half8& de_h, half8& de_l) {
    const long long s = sidx < a.M ? sidx : a.M - 1;
    if constexpr (IN_MODE == IN_ENC) {
        load_encoded_dir<SPLIT>(a.dirs_enc + s * 27, g, de_h, de_l);
    } else {
        double d[3];
        const double* src = IN_MODE == IN_RAYS ? a.rays_d + ((unsigned)s / (unsigned)a.N) * 3 : a.dirs + s * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = src[k];
        encode_dir<SPLIT, SPLIT>(d, g, de_h, de_l, (a.out_dirs_enc && sidx < a.M) ? a.out_dirs_enc + sidx * 27 : nullptr);
    }
}

}  // namespace tgtc
