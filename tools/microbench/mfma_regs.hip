// Microbenchmark 2: does the MFMA issue rate depend on operand register classes / number of distinct B fragments?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

// CLS: 0 = builtin (compiler picks), 1 = asm acc "v" B "v", 2 = asm acc "v" B "a", 3 = asm acc "a" B "v"
template <int CLS>
__device__ __forceinline__ void mf(float4v& acc, half8 a, half8 b) {
    if constexpr (CLS == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if constexpr (CLS == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if constexpr (CLS == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
    if constexpr (CLS == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int CLS, int NB>
__global__ void __launch_bounds__(256, 1) k(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    float4v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    half8 b[NB], a[4];
#pragma unroll
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(0.001f * (lane + j + i));
#pragma unroll
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.002f * (lane + 2 * j + i));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NB; ++m) mf<CLS>(acc[m & 3], a[(m >> 2) & 3], b[m]);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CLS, int NB>
void run(const char* name) {
    unsigned long long* d;
    float* sink;
    const int blocks = 256, threads = 256, iters = 2000;
    (void)hipMalloc(&d, blocks * 4 * 8);
    (void)hipMalloc(&sink, blocks * threads * 4);
    k<CLS, NB><<<blocks, threads>>>(d, sink, iters);
    k<CLS, NB><<<blocks, threads>>>(d, sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += v;
    printf("%-44s NB=%2d : %.2f cycles per MFMA\n", name, NB, s / h.size() / iters / NB);
    (void)hipFree(d), (void)hipFree(sink);
}

int main() {
    run<0, 4>("builtin");
    run<0, 32>("builtin");
    run<1, 4>("asm acc v, B v");
    run<1, 32>("asm acc v, B v");
    run<2, 32>("asm acc v, B a");
    run<3, 32>("asm acc a, B v");
    return 0;
}
