// The fp16mx hidden-layer loop in isolation: eight 256 -> 256 layers per pass over a persistent weight ring, 8 waves per CU
// -- once as the HIP loop of the product (mlp_mx.h dense_mx), once as the hand-placed streams of tools/gen_mx_asm.py.
// Checks that both produce the same bits and prints the matrix-pipe occupancy of each.
//
//   python tools/gen_mx_asm.py bench [dist=2 max_fill=3] > tools/microbench/mx_asm_bench.inc
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tgtc-style_amd/csrc -o tools/microbench/mx_layer tools/microbench/mx_layer.hip
//   tools/microbench/mx_layer [passes]
#define TGTC_ASM_DMA 1
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mx_asm.h"

namespace tgtc {

constexpr MxShape kBenchShape[8] = {{16, 2, 0}, {16, 2, 0}, {16, 2, 0}, {16, 2, 0}, {16, 2, 0}, {16, 2, 0}, {16, 2, 0}, {16, 2, 0}};
inline constexpr MxTable kBenchTable = mx_make_table(kBenchShape, kRingBytes);
constexpr int kNQ = kBenchTable.first[8];
constexpr int bench_units() {
    const int end = kBenchTable.off[kNQ - 1] + kMxKGroupBytes;
    return (end + kChunkBytes - 1) / kChunkBytes * kChunkBytes / 1024;
}
constexpr int kUnits = bench_units();
constexpr int kScaleOff = 10240;
constexpr int kBiasBytes = 16384;

#include "mx_asm_bench.inc"

using C = MlpCfg<8, 1, false, 4>;

__device__ __forceinline__ float hashf(unsigned x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return (float)(int)(x & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;   // [-1, 1)
}

template <int VAR>
__global__ void __launch_bounds__(512, 2) k(const char* stream, const char* bias, unsigned* out, unsigned long long* cyc, int passes) {
    __shared__ __attribute__((aligned(16))) char smem[C::RING_BYTES + kBiasBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    using Reader = MxReader<C, SingleStreamMap<kUnits>, kBenchTable, true>;
    Reader rd;
    const char* const streams[1] = {stream};
    rd.init(streams, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < kBiasBytes / (8 * 1024); ++j)
        lds_dma16(bias + (j * 8 + wave) * 1024 + lane * 16, smem + C::RING_BYTES + (j * 8 + wave) * 1024);
    wait_vmcnt<0>();
    rd.ring.next = rd.ring.src[0];
    rd.ring.persist_prologue();
    const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * g);
    const lds_cptr rs_lane = opaque((lds_cptr)smem + C::RING_BYTES + kScaleOff + 2 * n);
    unsigned check = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p) {
        MxAct<2> X, Y;
        half8 l16[4];
        static_for<16>([&](auto rt_) {
            constexpr int rt = decltype(rt_)::value;
            float4v a;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = hashf((unsigned)(((blockIdx.x * 8 + wave) * 64 + lane) * 64 + rt * 4 + r) + 977u * (unsigned)(p & 3));
            mx_store_act<rt, 0>(a, X, l16);
            mx_store_act<rt, 1>(a, X, l16);
        });
        rd.ring.next = rd.ring.src[0];
        rd.template enter<0, kNQ>();
        if constexpr (VAR == 0) {
            const half8 nop[1] = {};
            auto to_Y = [&](auto rt_, auto h_, const float4v& acc) { mx_store_act<decltype(rt_)::value, decltype(h_)::value>(acc, Y, l16); };
            auto to_X = [&](auto rt_, auto h_, const float4v& acc) { mx_store_act<decltype(rt_)::value, decltype(h_)::value>(acc, X, l16); };
            static_for<4>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                dense_mx<C, kBenchTable.first[2 * i], kNQ, 16, 2, 0, 512 * i>(rd, bias_lane, rs_lane, X, nop, nop, to_Y);
                dense_mx<C, kBenchTable.first[2 * i + 1], kNQ, 16, 2, 0, 512 * i + 256>(rd, bias_lane, rs_lane, Y, nop, nop, to_X);
            });
        } else {
            half8 keep[6] = {};
            mx_asm_bench_a(rd, bias_lane, rs_lane, X, Y, keep);
            mx_asm_bench_b(rd, bias_lane, rs_lane, Y, X, keep);
        }
        rd.template finish<kNQ>();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u4v h = __builtin_bit_cast(u4v, X.h[i]);
            check = check * 31u + (h[0] ^ (h[1] * 3u) ^ (h[2] * 5u) ^ (h[3] * 7u));
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int i = 0; i < 6; ++i) check = check * 31u + (X.h6[b][i] ^ (X.l6[b][i] * 3u));
            check = check * 31u + (unsigned)X.sc[b];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    wait_vmcnt<0>();
    out[blockIdx.x * 512 + threadIdx.x] = check;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

}  // namespace tgtc

using namespace tgtc;

static unsigned short f2h(float f) {
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}

template <int VAR>
static double run(const char* dstream, const char* dbias, unsigned* dout, unsigned long long* dcyc, int passes, std::vector<unsigned>& res, const char* name) {
    const int blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((k<VAR>), dim3(blocks), dim3(512), 0, 0, dstream, dbias, dout, dcyc, 4);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VAR>), dim3(blocks), dim3(512), 0, 0, dstream, dbias, dout, dcyc, passes);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); exit(1); }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    res.resize(blocks * 512);
    hipMemcpy(res.data(), dout, res.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double cyc_pass = sum / h.size() / passes;
    const double us_pass = ms * 1e3 / passes;
    // matrix-pipe work of one pass on one CU: 8 waves x 8 layers x (128 f16 16x16x32 + 64 fp6 16x16x128 at 1.08 of an f16 one)
    const double slots = 8.0 * 8 * (128 + 64 * 1.08);
    const double pipe = slots * 16384.0 / (us_pass * 1e-6) / (2.5e15 / 256);
    printf("%-10s %8.2f us / pass  %9.0f s_memtime units / pass / wave  (%.1f per MFMA)  matrix pipe %.3f\n", name, us_pass, cyc_pass, cyc_pass / 1536.0, pipe);
    return us_pass;
}

int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 200;
    const size_t stream_bytes = (size_t)(kUnits / 16 + 16 + 8) * kChunkBytes;
    std::vector<char> stream(stream_bytes, 0), bias(kBiasBytes, 0);
    unsigned s = 12345u;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return s; };
    auto gauss = [&] { float a = 0; for (int i = 0; i < 6; ++i) a += (float)(rnd() >> 8) * (1.0f / 16777216.0f); return (a - 3.0f) * 1.41f; };
    for (int q = 0; q < kNQ; ++q) {
        char* gp = stream.data() + kBenchTable.off[q];
        unsigned short* wh = reinterpret_cast<unsigned short*>(gp);
        for (int i = 0; i < 4 * 512; ++i) wh[i] = f2h(0.0884f * gauss());
        unsigned* w6 = reinterpret_cast<unsigned*>(gp + 4096);
        for (int i = 0; i < 3072 / 4; ++i) w6[i] = rnd() ^ (rnd() >> 11);
    }
    float* b = reinterpret_cast<float*>(bias.data());
    for (int i = 0; i < 8 * 256; ++i) b[i] = 0.05f * gauss() + 0.02f;
    unsigned short* re = reinterpret_cast<unsigned short*>(bias.data() + kScaleOff);
    for (int i = 0; i < 8 * 256; ++i) re[i] = (unsigned short)(123 | (109 << 8));
    char *dstream, *dbias;
    unsigned* dout;
    unsigned long long* dcyc;
    hipMalloc(&dstream, stream_bytes), hipMalloc(&dbias, kBiasBytes), hipMalloc(&dout, 256 * 512 * 4), hipMalloc(&dcyc, 256 * 8 * 8);
    hipMemcpy(dstream, stream.data(), stream_bytes, hipMemcpyHostToDevice);
    hipMemcpy(dbias, bias.data(), kBiasBytes, hipMemcpyHostToDevice);
    printf("groups %d, stream %d KiB (%d chunks), %d passes\n", kNQ, kUnits, kUnits / 16, passes);
    std::vector<unsigned> r0, r1;
    for (int rep = 0; rep < 3; ++rep) {
        run<0>(dstream, dbias, dout, dcyc, passes, r0, "hip");
        run<1>(dstream, dbias, dout, dcyc, passes, r1, "asm");
    }
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    unsigned nz = 0;
    for (auto v : r0) nz |= v;
    printf("checksums: %zu of %zu lanes differ (hip vs asm)%s\n", bad, r0.size(), nz ? "" : "  [all zero: suspicious]");
    return bad ? 2 : 0;
}
