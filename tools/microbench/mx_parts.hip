// Issue cost of the pieces of the fp16+fp6 inner loop, one wave per SIMD and two (s_memtime cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const int* seed, float* out, long long* clk, int iters) {
    const int l = threadIdx.x;
    half32 v;
    for (int i = 0; i < 32; ++i) v[i] = (_Float16)(float)((seed[(l + i) & 1023] & 255) * 0.01f);
    i8v A, B;
    for (int i = 0; i < 8; ++i) A[i] = seed[(l * 8 + i) & 1023], B[i] = seed[(l * 8 + i + 512) & 1023];
    int sa = 120 + (seed[l & 1023] & 7), sb = 121 + (seed[(l + 7) & 1023] & 7);
    asm volatile("" : "+v"(sa), "+v"(sb));
    float scale = 1.0f;
    asm volatile("" : "+v"(scale));
    float4v acc[4] = {};
    u6 r = {0, 0, 0, 0, 0, 0};
    unsigned x = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // independent cvt
                u6 t = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
                x ^= t[0] ^ t[5];
                v[u] += (_Float16)1;
            } else if (MODE == 1) {  // fp6 MFMA, VGPR scales, 4 independent chains
                acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[u & 3], 2, 2, 0, sa, 0, sb);
            } else if (MODE == 2) {  // fp6 MFMA, one dependent chain
                acc[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[0], 2, 2, 0, sa, 0, sb);
            } else if (MODE == 3) {  // f16 MFMA, one dependent chain
                half8 ha = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7), hb = __builtin_shufflevector(v, v, 8, 9, 10, 11, 12, 13, 14, 15);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[0], 0, 0, 0);
            } else if (MODE == 4) {  // the group pattern: M C M M C M on two chains
                half8 ha = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7), hb = __builtin_shufflevector(v, v, 8, 9, 10, 11, 12, 13, 14, 15);
                if (u == 1 || u == 4) acc[1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[1], 2, 2, 0, sa, 0, sb);
                else if (u < 6) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[0], 0, 0, 0);
            } else if (MODE == 6 || MODE == 7) {  // M C M M C M on ONE chain (6) / alternating between two chains (7: the 128-deep blocks of a row tile)
                half8 ha = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7), hb = __builtin_shufflevector(v, v, 8, 9, 10, 11, 12, 13, 14, 15);
                if (u < 6) {
#pragma unroll
                    for (int c = 0; c < (MODE == 7 ? 2 : 1); ++c) {
                        if (u == 1 || u == 4) acc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[c], 2, 2, 0, sa, 0, sb);
                        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[c], 0, 0, 0);
                    }
                }
            } else if (MODE == 5) {  // f16 MFMA, 4 independent chains
                half8 ha = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7), hb = __builtin_shufflevector(v, v, 8, 9, 10, 11, 12, 13, 14, 15);
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[u & 3], 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)x + r[0];
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + l] = s;
    if (l == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    std::vector<int> seed(1024);
    unsigned x = 12345;
    for (auto& v : seed) { x = x * 1664525u + 1013904223u; v = (int)x; }
    int* dseed; float* dout; long long* dclk;
    CK(hipMalloc(&dseed, 4096)); CK(hipMalloc(&dout, 1024 * 256 * 4)); CK(hipMalloc(&dclk, 1024 * 8));
    CK(hipMemcpy(dseed, seed.data(), 4096, hipMemcpyHostToDevice));
    const char* names[8] = {"cvt_pk32_fp6_f16", "fp6 mfma 4 chains", "fp6 mfma dependent", "f16 mfma dependent", "group M C M M C M (per 8 slots, 6 used)", "f16 mfma 4 chains",
                            "group M C M M C M on ONE chain (6 MFMA per 8 slots)", "the same on two alternating chains (12 MFMA per 8 slots)"};
    const int iters = 2000;
    for (int waves = 1; waves <= 2; ++waves)
        for (int mode = 0; mode < 8; ++mode) {
            const int blocks = 256 * waves;
            switch (mode) {
                case 0: k<0><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
                case 1: k<1><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
                case 2: k<2><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
                case 3: k<3><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
                case 4: k<4><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
                case 5: k<5><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
                case 6: k<6><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
                case 7: k<7><<<blocks, 256>>>(dseed, dout, dclk, iters); break;
            }
            CK(hipDeviceSynchronize());
            std::vector<long long> clk(blocks);
            CK(hipMemcpy(clk.data(), dclk, blocks * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (auto c : clk) sum += (double)c;
            printf("%d wave(s)/SIMD  %-42s %.1f cycles per loop slot (8 slots per iteration)\n", waves, names[mode], sum / blocks / iters / 8);
        }
    return 0;
}
