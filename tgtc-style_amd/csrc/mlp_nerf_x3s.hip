// The NeRF trunk + sigma head in TGTC_PREC_FP16X3 (the coarse pass of a render: 128 depths per ray, only the density is
// used) on TWO column tiles per wave, one wave per SIMD: a persistent kernel (one workgroup of four waves per CU, 128 samples per
// pass, the weight ring never drains between passes) whose pass is ONE generated instruction stream (tools/gen_x3_asm.py ->
// x3_asm_nerf.inc).  Same arithmetic per sample, in the same order, as nerf_mlp_kernel<CfgExact, IN_RAYS, false> (mlp_nerf.hip;
// reference models.py:95-103 inside :216-223): bit-identical densities, half the weight-side work per MFMA.
#define TGTC_ASM_DMA 1  // see mlp_core.h lds_dma16
#include "mlp_nerf_mx.h"

#include "mlp_layouts.h"

namespace tgtc {

#ifndef TGTC_X3S_INC   // (timing experiments build against streams generated with tools/gen_x3_asm.py abl_*=1)
#define TGTC_X3S_INC "x3_asm_nerf.inc"
#endif
#include TGTC_X3S_INC

using CfgX3s = MlpCfg<4, 2, true, 4>;

__global__ void __launch_bounds__(256, 1) nerf_x3s_kernel(NerfArgs a, long long n_pass) {
    using C = CfgX3s;
    using Ring = WeightStream<C, SingleStreamMap<NerfLayout::kFragsSigma>, true, true>;
    // the stream's ring: 128 KiB in chunks of kX3sChunkBytes (the generator's `chunk`), every wave moves 1/4 of a chunk
    constexpr int kChunk = kX3sChunkBytes, kSlots = C::RING_BYTES / kChunk, kLook = kSlots - 1, kPiece = kChunk / C::NWAVES;
    constexpr int kPad = ((NerfLayout::kFragsSigma * C::FRAG_BYTES + kChunk - 1) / kChunk + kSlots - 1) / kSlots * kSlots;
    static_assert(NerfLayout::kFragsSigma == kX3sFrags && kPad == kX3sPadChunks && C::RING_BYTES == 131072 && C::FRAG_BYTES == 2048 && kPiece % 4096 == 0,
                  "the generated stream was laid out for this ring");

    // ring | biases
    __shared__ __attribute__((aligned(16))) char smem[C::RING_BYTES + kNerfBiasBytes];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ring ring;
    {
        const int lane = threadIdx.x & 63;
        const char* const streams[1] = {a.stream};
        ring.init(streams, smem, wave, lane);
#pragma unroll
        for (int j = 0; j < kNerfBiasBytes / (C::NWAVES * 1024); ++j)
            lds_dma16(a.bias + (j * C::NWAVES + wave) * 1024 + lane * 16, smem + C::RING_BYTES + (j * C::NWAVES + wave) * 1024);
        // chunks 0 .. kLook-1 of the stream: the state every pass's entry expects (the first entry's counted wait also covers
        // the bias table: it is older than the chunks)
        ring.lds_wave = smem + wave * kPiece;
        ring.voff = wave * kPiece + lane * 16;
        static_for<kLook * (kPiece / 1024)>([&](auto i_) {
            constexpr int i = decltype(i_)::value, ch = i / (kPiece / 1024), j = i % (kPiece / 1024);
            lds_dma16_s<(j % 4) * 1024>(ring.src[0] + ch * kChunk + (j / 4) * 4096, ring.voff, ring.lds_wave + ch * kChunk + (j / 4) * 4096);
        });
    }

    for (long long p = blockIdx.x; p < n_pass; p += gridDim.x) {
        const long long s_wave = p * C::SAMPLES_PER_WG + wave * C::SAMPLES_PER_WAVE;
        half8 keep[2][4];
        {
            const int lane = fresh_lane_id();
            const int g = lane >> 4, n = lane & 15;
            double pos[2][3], dir[2][3];
            long long sidx[2];
            nerf_load_samples<2, IN_RAYS>(a, s_wave, n, pos, dir, sidx);
            half8 pe_h[2][2], pe_l[2][2], de_h[1][2], de_l[1][2];
            nerf_encode<2, true, false>(a, pos, dir, sidx, g, pe_h, pe_l, de_h, de_l);
#pragma unroll
            for (int c = 0; c < 2; ++c) keep[c][0] = pe_h[0][c], keep[c][1] = pe_h[1][c], keep[c][2] = pe_l[0][c], keep[c][3] = pe_l[1][c];
        }
        // per-lane addresses from a lane id read HERE: nothing but them and the loop's scalars lives across the stream
        const int fl = fresh_lane_id();
        ring.voff = wave * kPiece + fl * 16;
        ring.lane_lo = opaque((lds_cptr)smem + fl * 16);
        ring.lane_hi = opaque((lds_cptr)smem + 65536 + fl * 16);
        const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * (fl >> 4));
        float sigma[2];
        x3_asm_nerf_sigma_pass(ring, wave, bias_lane, keep, sigma);
        const int lane = fresh_lane_id();
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const long long s = s_wave + c * 16 + lane;
                if (s < a.M) a.sigma[s] = sigma[c];
            }
        }
    }
    wait_vmcnt<0>();   // the look-ahead of a pass that will not run: its LDS-DMA must have landed before the workgroup ends
}

int nerf_x3s_launch(const NerfArgs& a, hipStream_t st) {
    using C = CfgX3s;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        TGTC_HIP_CHECK(hipGetDevice(&dev));
        TGTC_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cus = n > 0 ? n : 256;
    }
    const long long n_pass = (a.M + C::SAMPLES_PER_WG - 1) / C::SAMPLES_PER_WG;
    nerf_x3s_kernel<<<(unsigned)(n_pass < cus ? n_pass : cus), C::NWAVES * 64, 0, st>>>(a, n_pass);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

}  // namespace tgtc
