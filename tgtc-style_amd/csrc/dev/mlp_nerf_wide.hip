// Fused positional encoding + NeRF MLP in the wide formulation (mlp_wide.h: v_mfma_f32_32x32x16_f16, four waves per
// workgroup = one per SIMD, 32 samples per wave).  Same contract as nerf_mlp_kernel (mlp_nerf.hip, reference
// models.py:63-117 / :182-223); used for the fp16x3 precision when the inputs are rays or points.
#include "mlp_core.h"
#include "mlp_layouts.h"
#include "mlp_nerf_front.h"
#include "mlp_pack.h"
#include "mlp_wide.h"

namespace tgtc {

template <class C, int IN_MODE, bool FULL>
__global__ void __launch_bounds__(256, 1) nerf_wide_kernel(NerfArgs a) {
    constexpr bool SPLIT = C::SPLIT;
    constexpr int NFRAG = FULL ? NerfLayoutW::kFragsFull : NerfLayoutW::kFragsSigma;
    __shared__ __attribute__((aligned(16))) char smem[C::RING_BYTES + kNerfBiasBytes];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long long sidx = (long long)blockIdx.x * C::SAMPLES_PER_WG + wave * C::SAMPLES_PER_WAVE + r;
    const long long s = sidx < a.M ? sidx : a.M - 1;  // tail: duplicate the last sample, stores are masked

    // ---- inputs (ordinary loads before any LDS-DMA is in flight)
    double pos[3];
    if constexpr (IN_MODE == IN_RAYS) {
        const long long ray = (unsigned)s / (unsigned)a.N;
        const double t = (double)a.ts[s];
#pragma unroll
        for (int k = 0; k < 3; ++k) pos[k] = a.rays_o[ray * 3 + k] + t * a.rays_d[ray * 3 + k];  // rendering.py:27 / utils.py:529
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) pos[k] = a.pts[s * 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(pos[k]));

    WeightStream<C, SingleStreamMap<NFRAG>> ws;
    const char* const streams[1] = {a.stream};
    ws.init(streams, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < kNerfBiasBytes / (C::NWAVES * 1024); ++j)
        lds_dma16(a.bias + (j * C::NWAVES + wave) * 1024 + lane * 16, smem + C::RING_BYTES + (j * C::NWAVES + wave) * 1024);
    ws.prologue();

    half8 pe_h[4], pe_l[4];
    encode_point_w<SPLIT, SPLIT>(pos, h, pe_h, pe_l);

    const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 64 * h);
    ws.start();
    nerf_chain_w<C, FULL>(
        ws, bias_lane, pe_h, pe_l,
        [&](half8 (&dh)[2], half8 (&dl)[2]) {
            double d[3];
            const double* src = IN_MODE == IN_RAYS ? a.rays_d + ((unsigned)s / (unsigned)a.N) * 3 : a.dirs + s * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] = src[k];
            encode_dir_w<SPLIT, SPLIT>(d, h, dh, dl);
        },
        [&](float sigma) {
            if (h == 0 && a.sigma && sidx < a.M) a.sigma[sidx] = sigma;
        },
        [&](auto rt_, auto u_, const float16v& acc) {
            constexpr int rt = decltype(rt_)::value, u = decltype(u_)::value;
            if (a.remap && sidx < a.M) {
                float* o = a.remap + sidx * 256 + 32 * rt + 8 * (u >> 1) + 4 * h + 2 * (u & 1);
                o[0] = relu(acc[2 * u]), o[1] = relu(acc[2 * u + 1]);
            }
        },
        [&](const float16v& acc) {
            if (h == 0 && a.rgb && sidx < a.M) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a.rgb[sidx * 3 + c] = 1.0f / (1.0f + expf(-acc[c]));
            }
        });
}

// ------------------------------------------------------------------------------------------------ host
std::vector<LayerSpec> nerf_specs_w(const tgtc_linear* l) {
    std::vector<LayerSpec> v;
    auto add = [&](int idx, std::vector<Seg> segs) {
        v.push_back(LayerSpec{l[idx].weight, l[idx].bias, l[idx].out_features, l[idx].in_features, std::move(segs)});
    };
    add(0, {{SEG_PE63, 0, 4}});
    for (int i = 1; i <= 4; ++i) add(i, {{SEG_ACT, 0, 16}});
    add(5, {{SEG_ACT, 63, 16}, {SEG_PE63, 0, 4}});  // reference column order: [pe(63) | h(256)]
    for (int i = 6; i <= 9; ++i) add(i, {{SEG_ACT, 0, 16}});  // L6, L7, sigma_layer, base_remap_layer
    add(10, {{SEG_ACT, 0, 16}, {SEG_PE27, 256, 2}});  // rgb_layers.0 on [remap(256) | dirs(27)]
    add(11, {{SEG_ACT, 0, 8}});                       // rgb_layers.1
    return v;
}

int wide_col(const Seg& s, int k, int h, int j) {
    switch (s.kind) {
        case SEG_ACT: return act_col_w(k, h, j);
        case SEG_PE63: return pe63_col_w(k, h, j);
        case SEG_PE27: return pe27_col_w(k, h, j);
        default: return -1;
    }
}

// [bias table 16 KiB][stream] of the wide order; appended to the handle's allocation by tgtc_nerf_create
int nerf_wide_pack(const tgtc_linear* layers, bool split, std::vector<char>& out) {
    PackedNet p = pack_layers_w(nerf_specs_w(layers), split, wide_col);
    if (p.n_frags != NerfLayoutW::kFragsFull || (int)p.bias.size() != NerfLayoutW::kBiasFloats)
        return fail(TGTC_ERR_UNSUPPORTED, "nerf_create: wide layout mismatch (%d frags, %zu bias)", p.n_frags, p.bias.size());
    for (int i = 0; i < 12; ++i)
        if (p.frag0[i] != NerfLayoutW::frag0(i) || p.bias0[i] != NerfLayoutW::bias0(i))
            return fail(TGTC_ERR_UNSUPPORTED, "nerf_create: wide layout mismatch at layer %d", i);
    out.assign(kNerfBiasBytes + p.stream.size() * sizeof(half_t), 0);
    std::memcpy(out.data(), p.bias.data(), p.bias.size() * sizeof(float));
    std::memcpy(out.data() + kNerfBiasBytes, p.stream.data(), p.stream.size() * sizeof(half_t));
    return TGTC_OK;
}

template <int IN_MODE, bool FULL>
static int launch_wide(NerfArgs a, hipStream_t st) {
    using C = WideCfg<true>;
    if (a.M >= 0x7fffffffLL) return fail(TGTC_ERR_UNSUPPORTED, "nerf: too many samples in one launch (%lld)", a.M);
    const long long nwg = (a.M + C::SAMPLES_PER_WG - 1) / C::SAMPLES_PER_WG;
    nerf_wide_kernel<C, IN_MODE, FULL><<<(unsigned)nwg, 256, 0, st>>>(a);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

// fp16x3 handles only; in_mode IN_RAYS or IN_PTS
int nerf_wide_launch(const tgtc_net* net, int in_mode, bool full, NerfArgs a, hipStream_t st) {
    a.bias = net->dev + net->wide_off;
    a.stream = net->dev + net->wide_off + kNerfBiasBytes;
    if (in_mode == IN_RAYS) return full ? launch_wide<IN_RAYS, true>(a, st) : launch_wide<IN_RAYS, false>(a, st);
    return full ? launch_wide<IN_PTS, true>(a, st) : launch_wide<IN_PTS, false>(a, st);
}

}  // namespace tgtc
