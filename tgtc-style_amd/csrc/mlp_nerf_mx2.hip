// The NeRF MLP (all twelve layers, or the trunk + sigma head of a coarse pass) in TGTC_PREC_FP16_FP6 on TWO column tiles per wave, one wave per SIMD: a persistent kernel (one
// workgroup of four waves per CU, 128 samples per pass, the weight ring never drains between passes) whose pass is ONE
// generated instruction stream (tools/gen_mx2_asm.py -> mx2_asm_nerf.inc).  Same arithmetic per sample, in the same order,
// as nerf_mx_kernel (mlp_nerf_mx.hip, reference models.py:63-117 inside :182-223): bit-identical outputs, half the LDS
// bytes per MFMA -- which is what bounds the one-tile loop (profiles/r4_kernel_variants.md).
#define TGTC_ASM_DMA 1  // see mlp_core.h lds_dma16
#include "mlp_nerf_mx.h"

#include "mlp_layouts.h"
#include "mlp_mx.h"
#include "mlp_nerf_mx_chain.h"

namespace tgtc {

#ifndef TGTC_MX2_INC   // (timing experiments build against streams generated with tools/gen_mx2_asm.py abl_*=1)
#define TGTC_MX2_INC "mx2_asm_nerf.inc"
#endif
#include TGTC_MX2_INC

using CfgMx2 = MlpCfg<4, 2, false, 4>;
#ifndef TGTC_MX2_ABL
#define TGTC_MX2_ABL 0
#endif

// FULL = false: the trunk + sigma head alone (the coarse pass of a render whose coarse network is fp16mx), rays in, densities out
template <int IN_MODE, bool FULL>
__global__ void __launch_bounds__(256, 1) nerf_mx2_kernel(NerfArgs a, long long n_pass) {
    using C = CfgMx2;
    constexpr int NUNITS = nerf_mx_units(FULL);
    using Reader = MxReader<C, SingleStreamMap<NUNITS>, kNerfMxTable, true>;
    // the stream's ring: 128 KiB in chunks of kMx2ChunkBytes (the generator's `chunk`), every wave moves 1/4 of a chunk
    constexpr int kChunk = kMx2ChunkBytes, kSlots = C::RING_BYTES / kChunk, kLook = kSlots - 1, kPiece = kChunk / C::NWAVES;
    constexpr int kPad = ((NUNITS * 1024 + kChunk - 1) / kChunk + kSlots - 1) / kSlots * kSlots;
    static_assert(kPad == (FULL ? kMx2PadChunks : kMx2SigmaPadChunks) && C::RING_BYTES == 131072 && kPiece % 4096 == 0,
                  "the generated stream was laid out for this ring");

    // ring | biases + row exponents
    __shared__ __attribute__((aligned(16))) char smem[C::RING_BYTES + kNerfBiasBytes];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Reader rd;
    {
        const int lane = threadIdx.x & 63;
        const char* const streams[1] = {a.stream};
        rd.init(streams, smem, wave, lane);
#pragma unroll
        for (int j = 0; j < kNerfBiasBytes / (C::NWAVES * 1024); ++j)
            lds_dma16(a.bias + (j * C::NWAVES + wave) * 1024 + lane * 16, smem + C::RING_BYTES + (j * C::NWAVES + wave) * 1024);
        // chunks 0 .. kLook-1 of the stream: the state every pass's entry expects (the first entry's counted wait also covers
        // the bias table: it is older than the chunks)
        rd.ring.lds_wave = smem + wave * kPiece;
        rd.ring.voff = wave * kPiece + lane * 16;
        static_for<kLook * (kPiece / 1024)>([&](auto i_) {
            constexpr int i = decltype(i_)::value, ch = i / (kPiece / 1024), j = i % (kPiece / 1024);
            lds_dma16_s<(j % 4) * 1024>(rd.ring.src[0] + ch * kChunk + (j / 4) * 4096, rd.ring.voff, rd.ring.lds_wave + ch * kChunk + (j / 4) * 4096);
        });
    }

    for (long long p = blockIdx.x; p < n_pass; p += gridDim.x) {
        const long long s_wave = p * C::SAMPLES_PER_WG + wave * C::SAMPLES_PER_WAVE;
        half8 keep[2][FULL ? 6 : 4];
        {
            const int lane = fresh_lane_id();
            const int g = lane >> 4, n = lane & 15;
            double pos[2][3], dir[2][3];
            long long sidx[2];
            half8 pe_h[2][2], pe_l[2][2], de_h[1][2], de_l[1][2];
#if TGTC_MX2_ABL & 1   // timing experiment: no sample loads, no encoding (results wrong by construction)
#pragma unroll
            for (int c = 0; c < 2; ++c) pe_h[0][c] = pe_h[1][c] = pe_l[0][c] = pe_l[1][c] = de_h[0][c] = de_l[0][c] = half8{(_Float16)(float)(g + n)};
#elif TGTC_MX2_ABL & 4   // timing experiment: the sample loads (and their wait) without the encoders' arithmetic
            nerf_load_samples<2, IN_MODE>(a, s_wave, n, pos, dir, sidx);
#pragma unroll
            for (int c = 0; c < 2; ++c)
                pe_h[0][c] = pe_h[1][c] = pe_l[0][c] = pe_l[1][c] = de_h[0][c] = de_l[0][c] =
                    half8{(_Float16)(float)(pos[c][0] + pos[c][1] + pos[c][2] + dir[c][0] + dir[c][1] + dir[c][2])};
#else
            nerf_load_samples<2, IN_MODE>(a, s_wave, n, pos, dir, sidx);
            if constexpr (IN_MODE == IN_ENC) nerf_load_encoded<2, true, FULL>(a, sidx, g, pe_h, pe_l, de_h, de_l);
            else nerf_encode<2, true, FULL>(a, pos, dir, sidx, g, pe_h, pe_l, de_h, de_l);
#endif
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                keep[c][0] = pe_h[0][c], keep[c][1] = pe_h[1][c], keep[c][2] = pe_l[0][c], keep[c][3] = pe_l[1][c];
                if constexpr (FULL) keep[c][4] = de_h[0][c], keep[c][5] = de_l[0][c];
            }
        }
        // per-lane addresses from a lane id read HERE: nothing but them and the loop's scalars lives across the stream
        rd.relane(smem, wave);
        const int fl = fresh_lane_id();
        rd.ring.voff = wave * kPiece + fl * 16;
        const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * (fl >> 4));
        const lds_cptr rs_lane = opaque((lds_cptr)smem + C::RING_BYTES + kNerfMxScaleOff + 2 * (fl & 15));
        float sigma[2], rgb[2][3] = {};
        if constexpr (FULL) mx2_asm_nerf_full_pass(rd, wave, bias_lane, rs_lane, keep, sigma, rgb);
        else mx2_asm_nerf_sigma_pass(rd, wave, bias_lane, rs_lane, keep, sigma);
        const int lane = fresh_lane_id();
        if (lane < 16 && !(TGTC_MX2_ABL & 2)) {   // (2: timing experiment without the output stores)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const long long s = s_wave + c * 16 + lane;
                if (s < a.M) {
                    if (a.sigma) a.sigma[s] = sigma[c];
                    if constexpr (FULL) {
                        if (a.rgb) {
#pragma unroll
                            for (int r = 0; r < 3; ++r) a.rgb[s * 3 + r] = 1.0f / (1.0f + expf(-rgb[c][r]));   // models.py:111
                        }
                    }
                }
            }
        }
    }
    wait_vmcnt<0>();   // the look-ahead of a pass that will not run: its LDS-DMA must have landed before the workgroup ends
}

int nerf_mx2_launch(int in_mode, bool full, const NerfArgs& a, hipStream_t st) {
    using C = CfgMx2;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        TGTC_HIP_CHECK(hipGetDevice(&dev));
        TGTC_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cus = n > 0 ? n : 256;
    }
    const long long n_pass = (a.M + C::SAMPLES_PER_WG - 1) / C::SAMPLES_PER_WG;
    const unsigned nwg = (unsigned)(n_pass < cus ? n_pass : cus);
    const dim3 block(C::NWAVES * 64);
    switch (in_mode * 2 + (full ? 1 : 0)) {
        case IN_RAYS * 2 + 0: nerf_mx2_kernel<IN_RAYS, false><<<nwg, block, 0, st>>>(a, n_pass); break;
        case IN_RAYS * 2 + 1: nerf_mx2_kernel<IN_RAYS, true><<<nwg, block, 0, st>>>(a, n_pass); break;
        case IN_PTS * 2 + 1: nerf_mx2_kernel<IN_PTS, true><<<nwg, block, 0, st>>>(a, n_pass); break;
        case IN_ENC * 2 + 1: nerf_mx2_kernel<IN_ENC, true><<<nwg, block, 0, st>>>(a, n_pass); break;
        default: return fail(TGTC_ERR_UNSUPPORTED, "nerf (fp16+fp6, two tiles): no kernel for input mode %d, full %d", in_mode, (int)full);
    }
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

}  // namespace tgtc
