// Probe of the gfx950 MX path used by the fp16+fp6 precision mode (tools only, not part of the library):
//   1. bit layout / scale semantics of v_cvt_scalef32_pk32_fp6_f16
//   2. operand lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3) operands, by one-hot probing
//   3. per-lane E8M0 scale operands
//   4. issue rate against v_mfma_f32_16x16x32_f16
// Build: hipcc --offload-arch=gfx950 -O3 mfma_fp6.hip -o mfma_fp6 ; run: ./mfma_fp6 outdir
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));             \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__device__ inline i8v widen(u6 r) {
    i8v o = {(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0};
    return o;
}

__global__ void cvt_probe(const _Float16* in, unsigned* out, float scale) {
    half32 v;
    for (int i = 0; i < 32; ++i) v[i] = in[threadIdx.x * 32 + i];
    u6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}

// block (t, b): A is one-hot (value 1) at cvt-input position t = lane*32 + elem; B holds 1 or 2 by bit b of its position
__global__ void onehot_probe(float* out, int swap) {
    const int t = blockIdx.x, b = blockIdx.y, l = threadIdx.x;
    half32 va, vb;
    for (int i = 0; i < 32; ++i) {
        va[i] = (l * 32 + i == t) ? (_Float16)1 : (_Float16)0;
        vb[i] = (((l * 32 + i) >> b) & 1) ? (_Float16)2 : (_Float16)1;
    }
    i8v A = widen(__builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(va, 1.0f));
    i8v B = widen(__builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vb, 1.0f));
    float4v acc = {0, 0, 0, 0};
    if (swap) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B, A, acc, 2, 2, 0, 127, 0, 127);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 2, 2, 0, 127, 0, 127);
    for (int i = 0; i < 4; ++i) out[((size_t)(t * gridDim.y + b)) * 256 + l * 4 + i] = acc[i];
}

// all-ones operands, per-lane scale bytes from tables
__global__ void scale_probe(const int* sa, const int* sb, float* out) {
    const int l = threadIdx.x;
    half32 v;
    for (int i = 0; i < 32; ++i) v[i] = (_Float16)1;
    i8v A = widen(__builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, 1.0f));
    float4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, A, acc, 2, 2, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}

// which byte of the scale VGPR does op_sel pick?  all-ones operands, scale register 0x8382807f:
// byte 0 -> 128, byte 1 -> 256, byte 2 -> 1024, byte 3 -> 2048
template <int OPA, int OPB>
__global__ void opsel_probe(float* out) {
    half32 v;
    for (int i = 0; i < 32; ++i) v[i] = (_Float16)1;
    i8v A = widen(__builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, 1.0f));
    float4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, A, acc, 2, 2, OPA, (int)0x8382807f, OPB, (int)0x8382807f);
    if (threadIdx.x == 0) out[OPA * 4 + OPB] = acc[0];
}

template <int MODE>
__global__ void __launch_bounds__(256) rate_probe(const int* seed, float* out, long long* clk, int iters) {
    const int l = threadIdx.x;
    i8v A, B;
    for (int i = 0; i < 8; ++i) A[i] = seed[(l * 8 + i) & 1023], B[i] = seed[(l * 8 + i + 512) & 1023];
    typedef int i4v __attribute__((ext_vector_type(4)));
    const i4v ia = {A[0] & 0x3bff3bff, A[1] & 0x3bff3bff, A[2] & 0x3bff3bff, A[3] & 0x3bff3bff};  // finite halves
    const i4v ib = {B[0] & 0x3bff3bff, B[1] & 0x3bff3bff, B[2] & 0x3bff3bff, B[3] & 0x3bff3bff};
    half8 ha = __builtin_bit_cast(half8, ia), hb = __builtin_bit_cast(half8, ib);
    float4v acc[4] = {};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[u & 3], 0, 0, 0);
            else acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[u & 3], 2, 2, 0, 120, 0, 120);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + l] = s;
    if (l == 0) clk[blockIdx.x * 2] = t1 - t0, clk[blockIdx.x * 2 + 1] = r1 - r0;
}

static void dump(const std::string& path, const void* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    fwrite(p, 1, n, f);
    fclose(f);
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    // 1. conversion
    {
        std::vector<_Float16> in(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 32; ++i) {
                float v = (float)((l * 32 + i) % 61) * 0.125f;            // every e2m3 magnitude and beyond
                if ((l * 32 + i) % 7 == 3) v = -v;
                if (l >= 32) v *= 0.37f;                                   // rounding cases
                in[l * 32 + i] = (_Float16)v;
            }
        _Float16* din;
        unsigned* dout;
        CK(hipMalloc(&din, in.size() * 2));
        CK(hipMalloc(&dout, 64 * 6 * 4));
        CK(hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice));
        dump(dir + "/cvt_in.bin", in.data(), in.size() * 2);
        const float scales[3] = {1.0f, 4.0f, 0.25f};
        for (int s = 0; s < 3; ++s) {
            cvt_probe<<<1, 64>>>(din, dout, scales[s]);
            std::vector<unsigned> o(64 * 6);
            CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
            dump(dir + "/cvt_out" + std::to_string(s) + ".bin", o.data(), o.size() * 4);
        }
    }
    // 2. one-hot lane maps
    for (int swap = 0; swap < 2; ++swap) {
        float* dout;
        const size_t n = (size_t)2048 * 11 * 256;
        CK(hipMalloc(&dout, n * 4));
        onehot_probe<<<dim3(2048, 11), 64>>>(dout, swap);
        std::vector<float> o(n);
        CK(hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost));
        dump(dir + "/onehot" + std::to_string(swap) + ".bin", o.data(), n * 4);
        CK(hipFree(dout));
    }
    // 3. scales
    {
        std::vector<int> sa(64), sb(64);
        for (int l = 0; l < 64; ++l) {
            sa[l] = (127 + (l % 16 == 3 ? 1 : 0) + (l / 16 == 2 ? 2 : 0)) | 0x55443300;  // junk in the upper bytes
            sb[l] = (127 + (l % 16 == 5 ? 3 : 0) - (l / 16 == 1 ? 1 : 0)) | 0x11223300;
        }
        int *dsa, *dsb;
        float* dout;
        CK(hipMalloc(&dsa, 256));
        CK(hipMalloc(&dsb, 256));
        CK(hipMalloc(&dout, 1024));
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        scale_probe<<<1, 64>>>(dsa, dsb, dout);
        std::vector<float> o(256);
        CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
        dump(dir + "/scale_out.bin", o.data(), 1024);
    }
    {
        float* dout;
        CK(hipMalloc(&dout, 64));
        CK(hipMemset(dout, 0, 64));
        opsel_probe<0, 0><<<1, 64>>>(dout);
        opsel_probe<1, 0><<<1, 64>>>(dout);
        opsel_probe<2, 0><<<1, 64>>>(dout);
        opsel_probe<3, 0><<<1, 64>>>(dout);
        opsel_probe<0, 1><<<1, 64>>>(dout);
        opsel_probe<0, 2><<<1, 64>>>(dout);
        opsel_probe<0, 3><<<1, 64>>>(dout);
        float o[16];
        CK(hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost));
        printf("op_sel a=0..3 (b=0): %g %g %g %g   expected 128 256 1024 2048 if op_sel picks byte a\n", o[0], o[4], o[8], o[12]);
        printf("op_sel b=1..3 (a=0): %g %g %g       expected 256 1024 2048\n", o[1], o[2], o[3]);
    }
    // 4. rate
    {
        std::vector<int> seed(1024);
        unsigned x = 12345;
        for (auto& v : seed) {
            x = x * 1664525u + 1013904223u;
            v = (int)x;
        }
        int* dseed;
        float* dout;
        long long* dclk;
        const int blocks = 256 * 2;  // 2 waves per SIMD
        CK(hipMalloc(&dseed, 4096));
        CK(hipMalloc(&dout, blocks * 256 * 4));
        CK(hipMalloc(&dclk, blocks * 16));
        CK(hipMemcpy(dseed, seed.data(), 4096, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int iters = 20000;
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) rate_probe<0><<<blocks, 256>>>(dseed, dout, dclk, iters);
                else rate_probe<1><<<blocks, 256>>>(dseed, dout, dclk, iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<long long> clk(blocks * 2);
                CK(hipMemcpy(clk.data(), dclk, blocks * 16, hipMemcpyDeviceToHost));
                const double cyc = (double)clk[0], real = (double)clk[1];
                const double n_mfma = (double)iters * 16;
                printf("mode %s rep %d: %.3f ms, %.2f cycles/MFMA/wave (x2 waves per SIMD), clock %.0f MHz, %.1f TFLOP/s\n",
                       mode ? "fp6 16x16x128" : "f16 16x16x32", rep, ms, cyc / n_mfma, cyc / real * 100.0,
                       n_mfma * blocks * 4 * 2.0 * 16 * 16 * (mode ? 128 : 32) / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
