// One wave per SIMD (512 registers per lane), the fp16mx group pattern on TWO column tiles (two accumulator chains):
//   M0a M0b C1a C1b M1a M1b M2a M2b C2a C2b M3a M3b     (M: v_mfma_f32_16x16x32_f16, C: v_mfma_scale_f32_16x16x128_f8f6f4 e2m3)
// with the weight operand (A) and / or the activation operand (B) in architectural or accumulator registers, and with the
// loop's other instruction classes mixed in.  Question behind it (profiles/r4_kernel_variants.md section 3): can the weights
// of a two-column-tile fp16mx loop live in AGPRs -- the only place 56 more registers exist -- without slowing the matrix pipe?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/agpr_mfma tools/microbench/agpr_mfma.hip && tools/microbench/agpr_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// registers: accumulators v[100:103] (a), v[104:107] (b); f16 A v[110:113] / a[10:13]; f16 B v[120:123] (a), v[124:127] (b) /
// a[20:23], a[24:27]; fp6 A v[130:135] / a[30:35]; fp6 B v[140:145], v[146:151] / a[40:45], a[46:51]; scales v160, v161
#define M(acc, A, B) "v_mfma_f32_16x16x32_f16 " acc ", " A ", " B ", " acc "\n\t"
#define C(acc, A, B) "v_mfma_scale_f32_16x16x128_f8f6f4 " acc ", " A ", " B ", " acc ", v160, v161 op_sel:[1,0,0] op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
#define ACA "v[100:103]"
#define ACB "v[104:107]"

template <int VAR>
__global__ void __launch_bounds__(256, 1) k(long long* clk, float* out, int iters) {
    // VAR bit 0: f16/fp6 A in AGPR; bit 1: fp6 B in AGPR; bit 2: f16 B in AGPR; bit 3: + one VALU per MFMA; bit 4: + one ds_read_b128 (into the
    // A operand's file) per two MFMAs; bit 5: + one v_cvt_scalef32_pk32_fp6_f16 per 48 MFMAs
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 96 KiB: one workgroup per CU, one wave per SIMD
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + threadIdx.x * 16;
    asm volatile(
        "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
        "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
        "v_mov_b32 v160, 0x7b\n\tv_mov_b32 v161, 0x7b\n\t"
        ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v160", "v161");
#define FA (VAR & 1 ? "a[10:13]" : "v[110:113]")
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define VALU(n) (VAR & 8 ? "v_max_i32_e32 v" #n ", 0, v" #n "\n\t" : "")
#define BODY(FA_, FB0, FB1, XA, XB0, XB1, LD0, LD1, LD2, LD3, LD4, LD5, V0, V1, V2, V3, V4, V5, V6, V7, V8, V9, V10, V11) \
        asm volatile(M(ACA, FA_, FB0) V0 M(ACB, FA_, FB1) V1 LD0 C(ACA, XA, XB0) V2 C(ACB, XA, XB1) V3 LD1 M(ACA, FA_, FB0) V4 M(ACB, FA_, FB1) V5 LD2 \
                     M(ACA, FA_, FB0) V6 M(ACB, FA_, FB1) V7 LD3 C(ACA, XA, XB0) V8 C(ACB, XA, XB1) V9 LD4 M(ACA, FA_, FB0) V10 M(ACB, FA_, FB1) V11 LD5 \
                     ::"v"(lds) : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v170", "v171", "v172", "v173", \
                        "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67")
#define VV(n) "v_max_i32_e32 v17" #n ", 0, v17" #n "\n\t"
#define NOV ""
#define LDV(o) "ds_read_b128 v[180:183], %0 offset:" #o "\n\t"
#define LDA(o) "ds_read_b128 a[60:63], %0 offset:" #o "\n\t"
        if constexpr ((VAR & 24) == 0) {
            if constexpr ((VAR & 7) == 0) BODY("v[110:113]", "v[120:123]", "v[124:127]", "v[130:135]", "v[140:145]", "v[146:151]", NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV);
            if constexpr ((VAR & 7) == 1) BODY("a[10:13]", "v[120:123]", "v[124:127]", "a[30:35]", "v[140:145]", "v[146:151]", NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV);
            if constexpr ((VAR & 7) == 3) BODY("a[10:13]", "v[120:123]", "v[124:127]", "a[30:35]", "a[40:45]", "a[46:51]", NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV);
            if constexpr ((VAR & 7) == 7) BODY("a[10:13]", "a[20:23]", "a[24:27]", "a[30:35]", "a[40:45]", "a[46:51]", NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV);
            if constexpr ((VAR & 7) == 2) BODY("v[110:113]", "v[120:123]", "v[124:127]", "v[130:135]", "a[40:45]", "a[46:51]", NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV, NOV);
        } else if constexpr ((VAR & 24) == 8) {     // + VALU
            if constexpr ((VAR & 7) == 0) BODY("v[110:113]", "v[120:123]", "v[124:127]", "v[130:135]", "v[140:145]", "v[146:151]", NOV, NOV, NOV, NOV, NOV, NOV, VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3));
            if constexpr ((VAR & 7) == 3) BODY("a[10:13]", "v[120:123]", "v[124:127]", "a[30:35]", "a[40:45]", "a[46:51]", NOV, NOV, NOV, NOV, NOV, NOV, VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3));
        } else if constexpr ((VAR & 24) == 24) {    // + VALU + LDS reads
            if constexpr ((VAR & 7) == 0) BODY("v[110:113]", "v[120:123]", "v[124:127]", "v[130:135]", "v[140:145]", "v[146:151]", LDV(0), LDV(4096), LDV(8192), LDV(12288), LDV(16384), LDV(20480), VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3));
            if constexpr ((VAR & 7) == 3) BODY("a[10:13]", "v[120:123]", "v[124:127]", "a[30:35]", "a[40:45]", "a[46:51]", LDA(0), LDA(4096), LDA(8192), LDA(12288), LDA(16384), LDA(20480), VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3), VV(0), VV(1), VV(2), VV(3));
        }
        if constexpr ((VAR & 32) != 0) {
            if ((it & 3) == 3) asm volatile("v_cvt_scalef32_pk32_fp6_f16 v[184:189], v[200:215], v160\n\t" ::: "v184", "v185", "v186", "v187", "v188", "v189");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r;
    asm volatile("v_mov_b32 %0, v100" : "=v"(r));
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// The same question for 32x32x16 MFMAs (8 passes, twice the flops per instruction): two alternating chains (accumulators
// v[100:115], v[116:131]), A in a[10:13], B in v[120:123] / v[124:127]; VAR bit 0: + two VALU per MFMA, bit 1: + one ds_read_b128
// (into AGPRs) per MFMA -- the per-flop instruction mix of k<27> above.
template <int VAR>
__global__ void __launch_bounds__(256, 1) k32(long long* clk, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + threadIdx.x * 16;
    asm volatile("v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\t"
                 "v_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
                 "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\t"
                 "v_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v132, 0\n\tv_mov_b32 v133, 0\n\tv_mov_b32 v134, 0\n\tv_mov_b32 v135, 0\n\t"
                 "v_mov_b32 v136, 0\n\tv_mov_b32 v137, 0\n\tv_mov_b32 v138, 0\n\tv_mov_b32 v139, 0\n\tv_mov_b32 v140, 0\n\tv_mov_b32 v141, 0\n\t"
                 "v_mov_b32 v142, 0\n\tv_mov_b32 v143, 0\n\t" ::: "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v170", "v171", "v172", "v173", "a60", "a61", "a62", "a63");
    const long long t0 = __builtin_amdgcn_s_memtime();
#define BIG(acc, B) "v_mfma_f32_32x32x16_f16 " acc ", a[10:13], " B ", " acc "\n\t"
#define VV2(a, b) "v_max_i32_e32 v17" #a ", 0, v17" #a "\n\tv_max_i32_e32 v17" #b ", 0, v17" #b "\n\t"
    for (int it = 0; it < iters; ++it) {
        if constexpr (VAR == 0)
            asm volatile(BIG("v[100:115]", "v[120:123]") BIG("v[132:147]", "v[124:127]") BIG("v[100:115]", "v[120:123]") BIG("v[132:147]", "v[124:127]")
                         BIG("v[100:115]", "v[120:123]") BIG("v[132:147]", "v[124:127]") ::"v"(lds) : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v170", "v171", "v172", "v173", "a60", "a61", "a62", "a63");
        else if constexpr (VAR == 1)
            asm volatile(BIG("v[100:115]", "v[120:123]") VV2(0, 1) BIG("v[132:147]", "v[124:127]") VV2(2, 3) BIG("v[100:115]", "v[120:123]") VV2(0, 1)
                         BIG("v[132:147]", "v[124:127]") VV2(2, 3) BIG("v[100:115]", "v[120:123]") VV2(0, 1) BIG("v[132:147]", "v[124:127]") VV2(2, 3) ::"v"(lds) : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v170", "v171", "v172", "v173", "a60", "a61", "a62", "a63");
        else
            asm volatile(BIG("v[100:115]", "v[120:123]") VV2(0, 1) LDA(0) BIG("v[132:147]", "v[124:127]") VV2(2, 3) LDA(4096) BIG("v[100:115]", "v[120:123]") VV2(0, 1) LDA(8192)
                         BIG("v[132:147]", "v[124:127]") VV2(2, 3) LDA(12288) BIG("v[100:115]", "v[120:123]") VV2(0, 1) LDA(16384) BIG("v[132:147]", "v[124:127]") VV2(2, 3) LDA(20480)
                         ::"v"(lds) : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v170", "v171", "v172", "v173", "a60", "a61", "a62", "a63");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r;
    asm volatile("v_mov_b32 %0, v100" : "=v"(r));
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int VAR>
static void run32(const char* name, long long* dclk, float* dout, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k32<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    hipLaunchKernelGGL((k32<VAR>), dim3(256), dim3(256), 98304, 0, dclk, dout, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k32<VAR>), dim3(256), dim3(256), 98304, 0, dclk, dout, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double slots = 4.0 * iters * 6 * 2.0;      // six 32x32x16 per iteration and wave = twelve 16x16x32-equivalents
    printf("%-44s %8.3f ms  matrix pipe %.3f\n", name, ms, slots * 16384.0 / (ms * 1e-3) / (2.5e15 / 256));
}

template <int VAR>
static void run(const char* name, long long* dclk, float* dout, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    hipLaunchKernelGGL((k<VAR>), dim3(256), dim3(256), 98304, 0, dclk, dout, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<VAR>), dim3(256), dim3(256), 98304, 0, dclk, dout, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(1024);
    CK(hipMemcpy(h.data(), dclk, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    // matrix-pipe work: per iteration and wave 8 f16 16x16x32 + 4 fp6 16x16x128 (1.08 of an f16 one); 4 waves per CU
    const double slots = 4.0 * iters * (8 + 4 * 1.08);
    const double pipe = slots * 16384.0 / (ms * 1e-3) / (2.5e15 / 256);
    printf("%-44s %8.3f ms  %6.1f s_memtime units per MFMA and wave   matrix pipe %.3f\n", name, ms, s / h.size() / iters / 12.0, pipe);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    long long* dclk;
    float* dout;
    CK(hipMalloc(&dclk, 1024 * 8));
    CK(hipMalloc(&dout, 256 * 256 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A VGPR, B VGPR", dclk, dout, iters);
        run<1>("A AGPR, B VGPR", dclk, dout, iters);
        run<2>("A VGPR, fp6 B AGPR", dclk, dout, iters);
        run<3>("A AGPR, fp6 B AGPR (plan)", dclk, dout, iters);
        run<7>("A AGPR, all B AGPR", dclk, dout, iters);
        run<8>("all VGPR + 1 VALU / MFMA", dclk, dout, iters);
        run<11>("plan + 1 VALU / MFMA", dclk, dout, iters);
        run<24>("all VGPR + VALU + ds_read_b128 / 2 MFMA", dclk, dout, iters);
        run<27>("plan + VALU + ds_read_b128 (to AGPR) / 2 MFMA", dclk, dout, iters);
        run<59>("plan + VALU + reads + cvt_scalef32 / 48 MFMA", dclk, dout, iters);
        run32<0>("32x32x16: two chains, A in AGPRs", dclk, dout, iters);
        run32<1>("32x32x16 + 2 VALU / MFMA", dclk, dout, iters);
        run32<2>("32x32x16 + 2 VALU + ds_read_b128 / MFMA", dclk, dout, iters);
    }
    return 0;
}
