// Does LDS -> VGPR return traffic slow the matrix pipe down?  8 waves per CU (2 per SIMD), each iteration issues
// NMF independent 16x16x32 f16 MFMAs and NRD ds_read_b128 (1 KiB per wave-instruction) whose data feed the NEXT
// iteration's MFMAs (double buffered, counted lgkmcnt waits since no LDS-DMA is visible to the compiler).
// Prints s_memtime cycles per iteration per wave; compare against NMF*16*2 (two waves share a SIMD's pipe) and
// NRD*8 waves*1 KiB / 256 B per clk.
// hipcc --offload-arch=gfx950 -O3 -o lds_mfma_mix lds_mfma_mix.hip && ./lds_mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const char* lds_cptr;
typedef __attribute__((address_space(3))) const half8* lds_h8;

template <int NMF, int NRD, int NCH, int BAR, int NVALU, int IVALU = 0, int EVERY = 1, int KIND = 0, int RING = 0>
__global__ void __launch_bounds__(512, 2) k(unsigned long long* out, float* sink, int iters, const char* wsrc) {
    __shared__ __attribute__((aligned(16))) char smem[131072];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 131072 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * (i & 1023);
    __syncthreads();
    unsigned a0 = (unsigned)(size_t)((lds_cptr)smem + lane * 16);
    asm volatile("" : "+v"(a0));
    lds_cptr base = (lds_cptr)(size_t)a0;
    float4v acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = float4v{0, 0, 0, 0};
    half8 b[4];
    for (int c = 0; c < 4; ++c)
        for (int j = 0; j < 8; ++j) b[c][j] = (_Float16)(0.01f * (lane + j + c));
    constexpr int NQ = NRD > 0 ? NRD : 1;
    half8 q[2][NQ];
    for (int j = 0; j < NQ; ++j) q[0][j] = q[1][j] = b[j & 3];
    float junk = 0.001f * lane;
    float jv[8];
    for (int v = 0; v < 8; ++v) jv[v] = 0.1f * v + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto timed = [&](auto late_) {
        constexpr bool late = decltype(late_)::value;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                const half8 a = NRD > 0 ? q[half][m % NQ] : b[(m + 1) & 3];
                acc[m % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[m & 3], acc[m % NCH], 0, 0, 0);
#ifdef BURST_READS
                if (NRD > 0 && m == 0) {  // all reads of the iteration in one burst behind the first MFMA
#pragma unroll
                    for (int j = 0; j < NRD; ++j) q[half ^ 1][j] = *(lds_h8)(base + ((it + half) & 7) * 8192 + j * 1024);
                }
#else
                if (NRD > 0 && !late && m < NRD)   // one read behind each of the first NRD MFMAs, into the other buffer
                    q[half ^ 1][m] = *(lds_h8)(base + ((it + half) & 7) * 8192 + m * 1024);
                if (NRD > 0 && late && m >= NMF - NRD)
                    q[half ^ 1][m - (NMF - NRD)] = *(lds_h8)(base + ((it + half) & 7) * 8192 + (m - (NMF - NRD)) * 1024);
#endif
                if (IVALU > 0 && m % EVERY == EVERY - 1) {   // IVALU independent VALU instructions behind every EVERY-th MFMA
#pragma unroll
                    for (int v = 0; v < IVALU; ++v) {
                        float& x = jv[(m * IVALU + v) & 7];
                        if (KIND == 0) x = x * 1.0001f + 0.5f;                                              // v_fma_f32
                        else if (KIND == 1) x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), it));   // v_max_i32
                        else if (KIND == 2) x = (float)(_Float16)x;                                          // v_cvt_f16_f32 + v_cvt_f32_f16
                        else if (KIND == 3) x = x - jv[(m + 1) & 7];                                         // v_sub_f32
                        else if (KIND == 4) asm volatile("v_mov_b32 %0, %0" : "+v"(x));                      // v_mov_b32
                    }
                }
            }
            if (NVALU > 0) {   // epilogue-like VALU work (max, cvt, sub) on values the MFMAs do not depend on
#pragma unroll
                for (int v = 0; v < NVALU; ++v) junk = fmaxf(junk * 1.0001f, (float)(_Float16)junk) - 0.5f;
            }
            if (RING) {   // the weight ring of the kernels: 1 KiB of LDS-DMA per wave and iteration, acquire every 2 iterations
                const char* gsrc = wsrc + (((it + half) & 63) * 8 + (threadIdx.x >> 6)) * 1024 + lane * 16;
                char* ldst = smem + 65536 + (((it + half) & 7) * 8 + (threadIdx.x >> 6)) * 1024;
                if (RING & 1)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
                if (RING & 2) {
                    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ldst);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(l) : "memory");
                }
                if ((RING & 4) && (RING & 8 ? (half == 1 && (it & 2)) : half == 1)) {   // bit 8: every 4 iterations instead of every 2
                    if (RING & 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            if (BAR) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    };
#ifdef STAGGER
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0) timed(std::true_type{});   // the second wave of every SIMD
    else timed(std::false_type{});
#else
    timed(std::false_type{});
#endif
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = junk;
    for (int v = 0; v < 8; ++v) s += jv[v];
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NMF, int NRD, int NCH = 8, int BAR = 0, int NVALU = 0, int IVALU = 0, int EVERY = 1, int KIND = 0, int RING = 0>
static void run(unsigned long long* dout, float* dsink) {
    static char* wsrc = nullptr;
    if (!wsrc) { hipMalloc(&wsrc, 1 << 20); hipMemset(wsrc, 0, 1 << 20); }
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL((k<NMF, NRD, NCH, BAR, NVALU, IVALU, EVERY, KIND, RING>), dim3(blocks), dim3(512), 0, 0, dout, dsink, iters, (const char*)wsrc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double cyc = sum / h.size() / iters;
    printf("NMF %2d NRD %d chains %d barrier %d bulk-valu %2d valu %d behind every %d. mfma kind %d ring %d : %7.1f cycles / iteration / wave   (MFMA pipe %4d, LDS array %4d)  pipe busy %.2f\n", NMF, NRD, NCH, BAR, NVALU, IVALU, EVERY, KIND, RING, cyc,
           NMF * 16 * 2, NRD * 8 * 4, NMF * 32.0 / cyc, NMF * 25.6 / cyc);
}

int main() {
    unsigned long long* dout;
    float* dsink;
    hipMalloc(&dout, 256 * 8 * 8);
    hipMalloc(&dsink, 256 * 512 * 4);
    run<8, 0>(dout, dsink);
    run<8, 2>(dout, dsink);
    run<8, 4>(dout, dsink);
    run<8, 6>(dout, dsink);
    run<8, 8>(dout, dsink);
    run<6, 7>(dout, dsink);   // fp16mx-like ratio (7 KiB per 6 MFMAs)
    run<12, 8>(dout, dsink);  // fp16x3-like ratio (8 KiB per 12 MFMAs)
    run<16, 4>(dout, dsink);  // fp16 NCT=2-like ratio (4 KiB per 16 MFMAs... per wave 8 MFMA-pairs)
    printf("dependent accumulator chains, as in the kernels:\n");
    run<12, 0, 1>(dout, dsink);
    run<12, 8, 1>(dout, dsink);  // fp16x3: one chain
    run<12, 8, 2>(dout, dsink);
    run<16, 0, 2>(dout, dsink);
    run<16, 4, 2>(dout, dsink);  // fp16 NCT=2: two chains
    run<6, 7, 2>(dout, dsink);   // fp16mx: two chains
    printf("plus a workgroup barrier per iteration / plus epilogue-like VALU work:\n");
    run<12, 8, 1, 1, 0>(dout, dsink);
    run<12, 8, 1, 0, 12>(dout, dsink);
    run<12, 8, 1, 0, 24>(dout, dsink);
    run<12, 8, 1, 1, 12>(dout, dsink);
    run<16, 4, 2, 1, 0>(dout, dsink);
    run<16, 4, 2, 0, 16>(dout, dsink);
    run<16, 4, 2, 1, 16>(dout, dsink);
    printf("independent VALU instructions interleaved behind every MFMA:\n");
    run<12, 8, 1, 0, 0, 1>(dout, dsink);
    run<12, 8, 1, 0, 0, 2>(dout, dsink);
    run<12, 8, 1, 0, 0, 3>(dout, dsink);
    run<12, 0, 1, 0, 0, 2>(dout, dsink);
    run<16, 4, 2, 0, 0, 1>(dout, dsink);
    run<16, 4, 2, 0, 0, 2>(dout, dsink);
    printf("one VALU instruction of each kind behind every MFMA (0 fma, 1 max_i32, 2 cvt f16 round trip = 2 instr, 3 sub, 4 mov), no LDS reads:\n");
    run<12, 0, 1, 0, 0, 1, 1, 0>(dout, dsink);
    run<12, 0, 1, 0, 0, 1, 1, 1>(dout, dsink);
    run<12, 0, 1, 0, 0, 1, 1, 2>(dout, dsink);
    run<12, 0, 1, 0, 0, 1, 1, 3>(dout, dsink);
    run<12, 0, 1, 0, 0, 1, 1, 4>(dout, dsink);
    printf("with the weight ring (LDS-DMA 1 KiB per wave-iteration, vmcnt wait + barrier every 2 iterations):\n");
    // ring bits: 1 = DMA through the builtin (visible to hipcc), 2 = DMA through inline asm, 4 = vmcnt wait + barrier
    run<12, 8, 1, 0, 0, 0, 1, 0, 5>(dout, dsink);
    run<12, 8, 1, 0, 0, 0, 1, 0, 6>(dout, dsink);
    run<12, 8, 1, 0, 0, 0, 1, 0, 1>(dout, dsink);
    run<12, 8, 1, 0, 0, 0, 1, 0, 2>(dout, dsink);
    run<12, 8, 1, 0, 0, 0, 1, 0, 4>(dout, dsink);
    run<12, 8, 1, 0, 0, 0, 1, 0, 12>(dout, dsink);
    run<12, 8, 1, 0, 0, 0, 1, 0, 13>(dout, dsink);
    run<16, 4, 2, 0, 0, 0, 1, 0, 5>(dout, dsink);
    run<16, 4, 2, 0, 0, 0, 1, 0, 6>(dout, dsink);
    run<16, 4, 2, 0, 0, 0, 1, 0, 2>(dout, dsink);
    run<16, 4, 2, 0, 0, 0, 1, 0, 4>(dout, dsink);
    printf("the same VALU count, clustered:\n");
    run<12, 8, 1, 0, 0, 3, 3>(dout, dsink);
    run<12, 8, 1, 0, 0, 6, 6>(dout, dsink);
    run<12, 8, 1, 0, 0, 12, 12>(dout, dsink);
    run<16, 4, 2, 0, 0, 4, 4>(dout, dsink);
    run<16, 4, 2, 0, 0, 8, 8>(dout, dsink);
    return 0;
}
