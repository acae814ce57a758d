// Microbenchmark: cycles per 16-MFMA group for the inner-loop shapes of the fused MLP kernel.
// hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const char* lds_cptr;
typedef __attribute__((address_space(3))) const half8* lds_h8;

template <int MODE, int NMF, int NRD>
__global__ void __launch_bounds__(256, 1) k(unsigned long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[131072];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 131072 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    unsigned a0 = (unsigned)(size_t)((lds_cptr)smem + lane * 16);
    asm volatile("" : "+v"(a0));
    lds_cptr base = (lds_cptr)(size_t)a0;
    float4v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    half8 b[4];
    for (int c = 0; c < 4; ++c)
        for (int j = 0; j < 8; ++j) b[c][j] = (_Float16)(0.01f * (lane + j + c));
    half8 q[2][4];
    for (int j = 0; j < NRD; ++j) q[0][j] = *(lds_h8)(base + j * 1024);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (MODE >= 1) {
#pragma unroll
                for (int j = 0; j < NRD; ++j) asm volatile("" ::"v"(q[half][j]));
#pragma unroll
                for (int j = 0; j < NRD; ++j) q[half ^ 1][j] = *(lds_h8)(base + ((it + half) & 7) * 4096 + j * 1024 + 32768);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                const half8 a = (MODE >= 1) ? q[half][m % NRD] : b[(m + 1) & 3];
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[m & 3], acc[m & 3], 0, 0, 0);
                if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NMF, int NRD>
void run(const char* name, int threads, int blocks) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, blocks * 4 * 8);
    hipMalloc(&sink, blocks * threads * 4);
    const int iters = 2000;
    k<MODE, NMF, NRD><<<blocks, threads>>>(d, sink, iters);
    k<MODE, NMF, NRD><<<blocks, threads>>>(d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    int n = 0;
    for (int i = 0; i < blocks; ++i)
        for (int w = 0; w < threads / 64; ++w) s += h[i * 4 + w], ++n;
    printf("%-34s threads %3d blocks %4d : %.1f cycles per group of %d MFMA (ideal %d)\n", name, threads, blocks,
           s / n / iters, NMF, NMF * 16);
    hipFree(d), hipFree(sink);
}

int main() {
    run<0, 16, 4>("mfma only", 64, 1);
    run<0, 16, 4>("mfma only", 256, 1);
    run<0, 16, 4>("mfma only", 256, 256);
    run<1, 16, 4>("mfma + 4 ds_read_b128 burst", 64, 1);
    run<1, 16, 4>("mfma + 4 ds_read_b128 burst", 256, 1);
    run<1, 16, 4>("mfma + 4 ds_read_b128 burst", 256, 256);
    run<1, 32, 4>("32 mfma + 4 reads", 256, 256);
    run<1, 16, 2>("16 mfma + 2 reads", 256, 256);
    return 0;
}
