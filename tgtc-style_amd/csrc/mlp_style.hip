// Stylised path: the concat MLP (reference models.py:120-147 StyleMLP_before_concat), the style MLP
// (models.py:149-180 StyleMLP_Wild_multilayers) and the fused per-sample chain of rendering.py:122-142
//   NeRF trunk -> sigma, base_remap;  concat MLP(pe, z) -> concat_features;
//   style MLP(cat(base_remap, concat_features), pe, mean(z)) -> rgb
// as ONE persistent kernel: 24 dense layers per sample, activations in registers, three packed weight
// streams (concat | NeRF trunk | style) walked back to back through the same LDS ring.
//
// Register budget: two 256-feature activation sets fit beside the accumulators; the chain needs a third
// (concat_features must survive the NeRF trunk, and style layer 0 reads remap + concat_features while
// producing its own 256 outputs).  The third set lives in a per-workgroup 128 KiB scratch slab in
// global memory (L2/MALL resident: 256 workgroups x 128 KiB), each lane storing and re-loading only its
// own 16-byte fragments -- no layout change, no cross-lane traffic, ~0.25 MB per 256 samples against
// ~3 MB of weight stream.
#include "mlp_core.h"
#include "mlp_layouts.h"
#include "mlp_pack.h"
#include "mlp_style_chain.h"

namespace tgtc {

struct StyledArgs {
    const char* nerf_bias;
    const char* nerf_stream;
    const char* pair_bias;
    const char* concat_stream;
    const char* style_stream;
    char* stash;  // gridDim.x * kStashBytesPerWG
    long long M;
    int N;
    const double* rays_o;
    const double* rays_d;
    const float* ts;
    const float* z;  // [R,32]
    float* rgb;
    float* sigma;
};

// ------------------------------------------------------------------------------------------------ fused
template <class C>
__global__ void __launch_bounds__(C::NWAVES * 64, C::NWAVES / 4) styled_rays_kernel(StyledArgs a) {
    constexpr int NCT = C::NCT;
    constexpr bool SPLIT = C::SPLIT;
    using Map = StyledMap<C>;
    using L = NerfLayout;
    __shared__ __attribute__((aligned(16))) char smem[kRingBytes + kNerfBiasBytes + kStylePairBiasBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    char* slab = a.stash + (size_t)blockIdx.x * kStashBytesPerWG + (size_t)tid * 16;   // this lane's 16-byte column

    WeightStream<C, Map> ws;
    const char* const streams[3] = {a.concat_stream, a.nerf_stream, a.style_stream};
    ws.init(streams, smem, wave, lane);
    // bias tables: loaded once per workgroup (LDS-DMA), visible after the first ring barrier
#pragma unroll
    for (int j = 0; j < kNerfBiasBytes / (C::NWAVES * 1024); ++j)
        __builtin_amdgcn_global_load_lds(TGTC_GPTR(a.nerf_bias + (j * C::NWAVES + wave) * 1024 + lane * 16),
                                         TGTC_LPTR(smem + kRingBytes + (j * C::NWAVES + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < kStylePairBiasBytes / (C::NWAVES * 1024); ++j)
        __builtin_amdgcn_global_load_lds(TGTC_GPTR(a.pair_bias + (j * C::NWAVES + wave) * 1024 + lane * 16),
                                         TGTC_LPTR(smem + kRingBytes + kNerfBiasBytes + (j * C::NWAVES + wave) * 1024), 16, 0, 0);
    const lds_cptr nerf_bias = opaque((lds_cptr)smem + kRingBytes + 16 * g);
    const lds_cptr pair_bias = opaque((lds_cptr)smem + kRingBytes + kNerfBiasBytes + 16 * g);

    const long long n_tiles = (a.M + C::SAMPLES_PER_WG - 1) / C::SAMPLES_PER_WG;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // ---- inputs
        const long long s_wave = tile * C::SAMPLES_PER_WG + wave * C::SAMPLES_PER_WAVE;
        double pos[NCT][3];
        long long sidx[NCT];
        float zsum[NCT];
        half8 z_h[NCT], z_l[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            long long s = s_wave + c * 16 + n;
            sidx[c] = s;
            if (s >= a.M) s = a.M - 1;
            const long long r = (unsigned)s / (unsigned)a.N;
            const double t = (double)a.ts[s];
#pragma unroll
            for (int k = 0; k < 3; ++k) pos[c][k] = a.rays_o[r * 3 + k] + t * a.rays_d[r * 3 + k];
            const float* zr = a.z + r * 32 + 8 * g;
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) part += zr[j];
            load_vec32<SPLIT>(a.z + r * 32, g, z_h[c], z_l[c]);
            zsum[c] = part;
            // retire the loads before any LDS-DMA is issued (their wait would drain the whole prefetch)
#pragma unroll
            for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(pos[c][k]));
            asm volatile("" : "+v"(zsum[c]), "+v"(z_h[c]));
            if constexpr (SPLIT) asm volatile("" : "+v"(z_l[c]));
        }
        // previous tile: every wave must be done with the ring before it is refilled
        __builtin_amdgcn_s_barrier();
        ws.prologue();

        half8 pe_h[2][NCT], pe_l[2][NCT], zb_h[NCT], zb_l[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            half8 h2[2], l2[2];
            encode_point<SPLIT, SPLIT>(pos[c], g, h2, l2, nullptr);
            pe_h[0][c] = h2[0], pe_h[1][c] = h2[1], pe_l[0][c] = l2[0], pe_l[1][c] = l2[1];
            // rendering.py:126: mean over the 32 latent channels, broadcast back to 32 (rendering.py:139)
            float zs = zsum[c];
            zs += __shfl_xor(zs, 16);
            zs += __shfl_xor(zs, 32);
            splat8<SPLIT>(zs * (1.0f / 32.0f), zb_h[c], zb_l[c]);
        }
        ws.start();

        half8 Xh[8][NCT], Xl[8][NCT], Yh[8][NCT], Yl[8][NCT];
        // ---- concat MLP -> Y, parked in the scratch slab while the trunk runs
        concat_mlp<C, Map::F_CONCAT, 0>(ws, pair_bias, pe_h, pe_l, z_h, z_l, Xh, Xl, Yh, Yl);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int c = 0; c < NCT; ++c) stash_store<C>(slab, ks, c, Yh[ks][c], Yl[ks][c]);

        // ---- NeRF trunk (models.py:95-101)
        auto to_Y = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
            store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][c], Yl[rt / 2][c]);
        };
        auto to_X = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
            store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][c], Xl[rt / 2][c]);
        };
        constexpr int FN = Map::F_NERF;
        dense_layer<C, FN + L::frag0(0), 2, 16, L::bias0(0)>(ws, nerf_bias, pe_h, pe_l, to_Y);
        dense_layer<C, FN + L::frag0(1), 8, 16, L::bias0(1)>(ws, nerf_bias, Yh, Yl, to_X);
        dense_layer<C, FN + L::frag0(2), 8, 16, L::bias0(2)>(ws, nerf_bias, Xh, Xl, to_Y);
        dense_layer<C, FN + L::frag0(3), 8, 16, L::bias0(3)>(ws, nerf_bias, Yh, Yl, to_X);
        dense_layer<C, FN + L::frag0(4), 8, 16, L::bias0(4)>(ws, nerf_bias, Xh, Xl, to_Y);
        {
            half8 Bh[10][NCT], Bl[10][NCT];
#pragma unroll
            for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Yh[k], Yl[k]);
            append<C>(Bh, Bl, 8, pe_h[0], pe_l[0]);
            append<C>(Bh, Bl, 9, pe_h[1], pe_l[1]);
            dense_layer<C, FN + L::frag0(5), 10, 16, L::bias0(5)>(ws, nerf_bias, Bh, Bl, to_X);
        }
        dense_layer<C, FN + L::frag0(6), 8, 16, L::bias0(6)>(ws, nerf_bias, Xh, Xl, to_Y);
        dense_layer<C, FN + L::frag0(7), 8, 16, L::bias0(7)>(ws, nerf_bias, Yh, Yl, to_X);
        dense_layer<C, FN + L::frag0(8), 8, 1, L::bias0(8)>(ws, nerf_bias, Xh, Xl, [&](auto, auto c_, auto h_, const float4v& acc) {
            constexpr int c = decltype(c_)::value;
            if constexpr (decltype(h_)::value == 0)
                if (g == 0 && a.sigma && sidx[c] < a.M) a.sigma[sidx[c]] = acc[0];
        });
        dense_layer<C, FN + L::frag0(9), 8, 16, L::bias0(9)>(ws, nerf_bias, Xh, Xl, to_Y);  // base_remap -> Y
        ws.template skip<FN + kTrunkFrags, Map::GAP>();

        // ---- style layer 0 on [remap (Y) | concat_features (slab -> X) | pe | mean z]; outputs stream to the slab
        stash_load<C>(slab, Xh, Xl);
        {
            half8 Bh[19][NCT], Bl[19][NCT];
#pragma unroll
            for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Yh[k], Yl[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, 8 + k, Xh[k], Xl[k]);
            append<C>(Bh, Bl, 16, pe_h[0], pe_l[0]);
            append<C>(Bh, Bl, 17, pe_h[1], pe_l[1]);
            append<C>(Bh, Bl, 18, zb_h, zb_l);
            half8 Th[NCT], Tl[NCT];
            dense_layer<C, Map::F_STYLE + style_frag0(0), 19, 16, kConcatBiasFloats + style_bias0(0)>(
                ws, pair_bias, Bh, Bl, [&](auto rt_, auto c_, auto h_, const float4v& acc) {
                    constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value, hf = decltype(h_)::value;
                    store_act<C, rt, hf>(acc, Th[c], Tl[c]);
                    if constexpr ((rt & 1) && hf == 1) stash_store<C>(slab, rt / 2, c, Th[c], Tl[c]);
                });
        }
        stash_load<C>(slab, Xh, Xl);
        // ---- style layers 1..7 -> rgb (models.py:172-179)
        style_tail<C, Map::F_STYLE, kConcatBiasFloats>(ws, pair_bias, pe_h, pe_l, zb_h, zb_l, Xh, Xl, Yh, Yl,
                                                       [&](auto c_, auto h_, const float4v& acc) {
                                                           constexpr int c = decltype(c_)::value, hf = decltype(h_)::value;
                                                           if (g == 0 && a.rgb && sidx[c] < a.M) {
#pragma unroll
                                                               for (int r = 2 * hf; r < (hf ? 3 : 2); ++r)
                                                                   a.rgb[sidx[c] * 3 + r] = 1.0f / (1.0f + expf(-acc[r]));
                                                           }
                                                       });
    }
}

// ------------------------------------------------------------------------------------------------ granular
struct ConcatArgs {
    const char* pair_bias;
    const char* concat_stream;
    long long M;
    const float* x;       // [M,63]
    const float* latent;  // [M,32]
    float* out;           // [M,256]
};

template <class C>
__global__ void __launch_bounds__(C::NWAVES * 64, C::NWAVES / 4) concat_kernel(ConcatArgs a) {
    constexpr int NCT = C::NCT;
    constexpr bool SPLIT = C::SPLIT;
    __shared__ __attribute__((aligned(16))) char smem[kRingBytes + kStylePairBiasBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    const long long s_wave = (long long)blockIdx.x * C::SAMPLES_PER_WG + wave * C::SAMPLES_PER_WAVE;
    half8 pe_h[2][NCT], pe_l[2][NCT], z_h[NCT], z_l[NCT];
    long long sidx[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        sidx[c] = s_wave + c * 16 + n;
        const long long s = sidx[c] < a.M ? sidx[c] : a.M - 1;
        half8 h2[2], l2[2];
        load_encoded_point<SPLIT>(a.x + s * 63, g, h2, l2);
        pe_h[0][c] = h2[0], pe_h[1][c] = h2[1], pe_l[0][c] = l2[0], pe_l[1][c] = l2[1];
        load_vec32<SPLIT>(a.latent + s * 32, g, z_h[c], z_l[c]);
    }
    WeightStream<C, SingleStreamMap<kConcatFrags>> ws;
    const char* const streams[1] = {a.concat_stream};
    ws.init(streams, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < kStylePairBiasBytes / (C::NWAVES * 1024); ++j)
        __builtin_amdgcn_global_load_lds(TGTC_GPTR(a.pair_bias + (j * C::NWAVES + wave) * 1024 + lane * 16),
                                         TGTC_LPTR(smem + kRingBytes + (j * C::NWAVES + wave) * 1024), 16, 0, 0);
    ws.prologue();
    const lds_cptr bias_lane = opaque((lds_cptr)smem + kRingBytes + 16 * g);
    ws.start();
    half8 Xh[8][NCT], Xl[8][NCT], Yh[8][NCT], Yl[8][NCT];
    // layers 0..3 through the shared helper would also run layer 4 into registers; the granular op wants
    // the fp32 outputs of layer 4, so it is spelled out here
    auto to_Y = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][c], Yl[rt / 2][c]);
    };
    auto to_X = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][c], Xl[rt / 2][c]);
    };
    {
        half8 Bh[3][NCT], Bl[3][NCT];
        append<C>(Bh, Bl, 0, pe_h[0], pe_l[0]);
        append<C>(Bh, Bl, 1, pe_h[1], pe_l[1]);
        append<C>(Bh, Bl, 2, z_h, z_l);
        dense_layer<C, concat_frag0(0), 3, 16, 0>(ws, bias_lane, Bh, Bl, to_Y);
    }
    auto hidden = [&](auto layer_, const half8 (&Ah)[8][NCT], const half8 (&Al)[8][NCT], auto&& epi) {
        constexpr int l = decltype(layer_)::value;
        half8 Bh[9][NCT], Bl[9][NCT];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Ah[k], Al[k]);
        append<C>(Bh, Bl, 8, z_h, z_l);
        dense_layer<C, concat_frag0(l), 9, 16, 256 * l>(ws, bias_lane, Bh, Bl, epi);
    };
    hidden(ic<1>{}, Yh, Yl, to_X);
    hidden(ic<2>{}, Xh, Xl, to_Y);
    hidden(ic<3>{}, Yh, Yl, to_X);
    {
        half8 Bh[11][NCT], Bl[11][NCT];
#pragma unroll
        for (int k = 0; k < 8; ++k) append<C>(Bh, Bl, k, Xh[k], Xl[k]);
        append<C>(Bh, Bl, 8, z_h, z_l);
        append<C>(Bh, Bl, 9, pe_h[0], pe_l[0]);
        append<C>(Bh, Bl, 10, pe_h[1], pe_l[1]);
        dense_layer<C, concat_frag0(4), 11, 16, 256 * 4>(ws, bias_lane, Bh, Bl, [&](auto rt_, auto c_, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value, hf = decltype(h_)::value;
            if (sidx[c] < a.M) {
                float* o = a.out + sidx[c] * 256 + 16 * rt + 4 * g + 2 * hf;
                o[0] = relu(acc[2 * hf]), o[1] = relu(acc[2 * hf + 1]);
            }
        });
    }
}

struct StyleArgs {
    const char* pair_bias;
    const char* style_stream;
    long long M;
    const float* x;         // [M,63]
    const float* concated;  // [M,512]
    const float* latent;    // [M,32]
    float* rgb;             // [M,3]
};

template <class C>
__global__ void __launch_bounds__(C::NWAVES * 64, C::NWAVES / 4) style_kernel(StyleArgs a) {
    constexpr int NCT = C::NCT;
    constexpr bool SPLIT = C::SPLIT;
    __shared__ __attribute__((aligned(16))) char smem[kRingBytes + kStylePairBiasBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    const long long s_wave = (long long)blockIdx.x * C::SAMPLES_PER_WG + wave * C::SAMPLES_PER_WAVE;
    half8 pe_h[2][NCT], pe_l[2][NCT], z_h[NCT], z_l[NCT];
    half8 Bh[19][NCT], Bl[19][NCT];
    long long sidx[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        sidx[c] = s_wave + c * 16 + n;
        const long long s = sidx[c] < a.M ? sidx[c] : a.M - 1;
        half8 h2[2], l2[2];
        load_encoded_point<SPLIT>(a.x + s * 63, g, h2, l2);
        pe_h[0][c] = h2[0], pe_h[1][c] = h2[1], pe_l[0][c] = l2[0], pe_l[1][c] = l2[1];
        load_vec32<SPLIT>(a.latent + s * 32, g, z_h[c], z_l[c]);
        // concated [512] -> 16 k-steps in the accumulator-tile k order (act_col)
        const float* cc = a.concated + s * 512;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = cc[(ks >> 3) * 256 + act_col(ks & 7, g, j)];
                const half_t h = (half_t)v;
                Bh[ks][c][j] = h;
                if constexpr (SPLIT) Bl[ks][c][j] = (half_t)(v - (float)h);
            }
        }
    }
    WeightStream<C, SingleStreamMap<kStyleFrags>> ws;
    const char* const streams[1] = {a.style_stream};
    ws.init(streams, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < kStylePairBiasBytes / (C::NWAVES * 1024); ++j)
        __builtin_amdgcn_global_load_lds(TGTC_GPTR(a.pair_bias + (j * C::NWAVES + wave) * 1024 + lane * 16),
                                         TGTC_LPTR(smem + kRingBytes + (j * C::NWAVES + wave) * 1024), 16, 0, 0);
    ws.prologue();
    const lds_cptr bias_lane = opaque((lds_cptr)smem + kRingBytes + 16 * g);
    ws.start();
    half8 Xh[8][NCT], Xl[8][NCT], Yh[8][NCT], Yl[8][NCT];
    append<C>(Bh, Bl, 16, pe_h[0], pe_l[0]);
    append<C>(Bh, Bl, 17, pe_h[1], pe_l[1]);
    append<C>(Bh, Bl, 18, z_h, z_l);
    dense_layer<C, style_frag0(0), 19, 16, kConcatBiasFloats + style_bias0(0)>(
        ws, bias_lane, Bh, Bl, [&](auto rt_, auto c_, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value;
            store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][c], Xl[rt / 2][c]);
        });
    style_tail<C, 0, kConcatBiasFloats>(ws, bias_lane, pe_h, pe_l, z_h, z_l, Xh, Xl, Yh, Yl, [&](auto c_, auto h_, const float4v& acc) {
        constexpr int c = decltype(c_)::value, hf = decltype(h_)::value;
        if (g == 0 && sidx[c] < a.M) {
#pragma unroll
            for (int r = 2 * hf; r < (hf ? 3 : 2); ++r) a.rgb[sidx[c] * 3 + r] = 1.0f / (1.0f + expf(-acc[r]));
        }
    });
}

// ------------------------------------------------------------------------------------------------ host
static std::vector<LayerSpec> concat_specs(const tgtc_linear* l) {
    std::vector<LayerSpec> v;
    auto add = [&](int i, std::vector<Seg> segs) {
        v.push_back(LayerSpec{l[i].weight, l[i].bias, l[i].out_features, l[i].in_features, std::move(segs)});
    };
    add(0, {{SEG_PE63, 0, 2}, {SEG_VEC32, 63, 1}});                       // cat(x, latent)            models.py:141
    for (int i = 1; i <= 3; ++i) add(i, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}});  // cat(h, latent)
    add(4, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}, {SEG_PE63, 288, 2}});   // cat(h, latent, x)         models.py:142-143
    return v;
}
static std::vector<LayerSpec> style_specs(const tgtc_linear* l) {
    std::vector<LayerSpec> v;
    auto add = [&](int i, std::vector<Seg> segs) {
        v.push_back(LayerSpec{l[i].weight, l[i].bias, l[i].out_features, l[i].in_features, std::move(segs)});
    };
    // cat(concated(512) = [base_remap | concat_features], x, latent)                                   models.py:169-172
    add(0, {{SEG_ACT, 0, 8}, {SEG_ACT, 256, 8}, {SEG_PE63, 512, 2}, {SEG_VEC32, 575, 1}});
    for (int i = 1; i <= 3; ++i) add(i, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}});
    add(4, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}, {SEG_PE63, 288, 2}});
    add(5, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}});
    add(6, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}});
    add(7, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}});                       // -> 3, sigmoid                models.py:177-179
    return v;
}

using CfgFast = MlpCfg<8, 2, false, 4>;      // same geometry as the NeRF kernels (mlp_nerf.hip)
using CfgExact = MlpCfg<8, 1, true, 4>;
using CfgFastNarrow = MlpCfg<4, 2, false, 4>;   // granular style MLP: 512 extra input features live in registers
using CfgExactNarrow = MlpCfg<4, 1, true, 4>;

// Launch wrappers.  The fp16 instances are compiled in a translation unit of their own (this source with
// -DTGTC_TU_FP16_ONLY) so that the two big kernel sets build in parallel.
template <class C>
void launch_styled_rays(unsigned grid, const StyledArgs& a, hipStream_t st) {
    styled_rays_kernel<C><<<grid, C::NWAVES * 64, 0, st>>>(a);
}
template <class C>
void launch_concat(const ConcatArgs& a, hipStream_t st) {
    concat_kernel<C><<<(unsigned)((a.M + C::SAMPLES_PER_WG - 1) / C::SAMPLES_PER_WG), C::NWAVES * 64, 0, st>>>(a);
}
template <class C>
void launch_style(const StyleArgs& a, hipStream_t st) {
    style_kernel<C><<<(unsigned)((a.M + C::SAMPLES_PER_WG - 1) / C::SAMPLES_PER_WG), C::NWAVES * 64, 0, st>>>(a);
}
#define TGTC_STYLE_FP16_INSTANCES(PREFIX)                                                        \
    PREFIX template void launch_styled_rays<CfgFast>(unsigned, const StyledArgs&, hipStream_t); \
    PREFIX template void launch_concat<CfgFast>(const ConcatArgs&, hipStream_t);                \
    PREFIX template void launch_style<CfgFastNarrow>(const StyleArgs&, hipStream_t);
#ifdef TGTC_TU_FP16_ONLY
TGTC_STYLE_FP16_INSTANCES()
}  // namespace tgtc
#else
TGTC_STYLE_FP16_INSTANCES(extern)

int styled_forward_rays_impl(const tgtc_net* nerf, const tgtc_net* style, const double* rays_o, const double* rays_d,
                             const float* ts, const float* z, int64_t R, int N, float* rgb, float* sigma,
                             hipStream_t st) {
    if (nerf->precision != style->precision)
        return fail(TGTC_ERR_ARG, "styled_forward_rays: NeRF and style nets were packed with different precisions");
    StyledArgs a{};
    a.nerf_bias = nerf->dev, a.nerf_stream = nerf->dev + nerf->bias_bytes;
    a.pair_bias = style->dev, a.concat_stream = style->dev + style->bias_bytes;
    a.style_stream = style->dev + style->stream2_off, a.stash = style->dev + style->stash_off;
    a.M = R * (int64_t)N, a.N = N, a.rays_o = rays_o, a.rays_d = rays_d, a.ts = ts, a.z = z, a.rgb = rgb, a.sigma = sigma;
    if (a.M >= 0x7fffffffLL) return fail(TGTC_ERR_UNSUPPORTED, "styled_forward_rays: too many samples in one launch");
    if (nerf->precision == TGTC_PREC_FP16) {
        const long long tiles = (a.M + CfgFast::SAMPLES_PER_WG - 1) / CfgFast::SAMPLES_PER_WG;
        launch_styled_rays<CfgFast>((unsigned)(tiles < style->n_wg ? tiles : style->n_wg), a, st);
    } else {
        const long long tiles = (a.M + CfgExact::SAMPLES_PER_WG - 1) / CfgExact::SAMPLES_PER_WG;
        launch_styled_rays<CfgExact>((unsigned)(tiles < style->n_wg ? tiles : style->n_wg), a, st);
    }
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

}  // namespace tgtc

using namespace tgtc;

extern "C" int tgtc_style_create(const tgtc_linear* concat_layers, int n_concat, const tgtc_linear* style_layers,
                                 int n_style, int precision, tgtc_net** out) {
    TGTC_REQUIRE(out && (concat_layers || style_layers), "style_create: null argument");
    if (precision == TGTC_PREC_FP16_FP6)
        return fail(TGTC_ERR_UNSUPPORTED, "style_create: TGTC_PREC_FP16_FP6 is implemented for the NeRF nets only (use FP16X3 or FP16)");
    TGTC_REQUIRE(precision == TGTC_PREC_FP16 || precision == TGTC_PREC_FP16X3, "style_create: unknown precision %d", precision);
    static const int want_c[5][2] = {{256, 95}, {256, 288}, {256, 288}, {256, 288}, {256, 351}};
    static const int want_s[8][2] = {{256, 607}, {256, 288}, {256, 288}, {256, 288}, {256, 351}, {256, 288}, {256, 288}, {3, 288}};
    // either net may be absent (n == 0): it is packed as zeros, for callers that only run the other one
    std::vector<float> zeros(256 * 607, 0.0f);
    tgtc_linear zc[5], zs[8];
    for (int i = 0; i < 5; ++i) zc[i] = tgtc_linear{zeros.data(), zeros.data(), want_c[i][0], want_c[i][1]};
    for (int i = 0; i < 8; ++i) zs[i] = tgtc_linear{zeros.data(), zeros.data(), want_s[i][0], want_s[i][1]};
    if (!concat_layers || n_concat == 0) concat_layers = zc, n_concat = 5;
    if (!style_layers || n_style == 0) style_layers = zs, n_style = 8;
    if (n_concat != 5 || n_style != 8)
        return fail(TGTC_ERR_UNSUPPORTED, "style_create: expected 5 concat + 8 style linears (style_D=8), got %d + %d", n_concat, n_style);
    for (int i = 0; i < 5; ++i)
        if (!concat_layers[i].weight || !concat_layers[i].bias || concat_layers[i].out_features != want_c[i][0] ||
            concat_layers[i].in_features != want_c[i][1])
            return fail(TGTC_ERR_UNSUPPORTED, "style_create: concat layer %d is %dx%d, kernels are built for %dx%d", i,
                        concat_layers[i].out_features, concat_layers[i].in_features, want_c[i][0], want_c[i][1]);
    for (int i = 0; i < 8; ++i)
        if (!style_layers[i].weight || !style_layers[i].bias || style_layers[i].out_features != want_s[i][0] ||
            style_layers[i].in_features != want_s[i][1])
            return fail(TGTC_ERR_UNSUPPORTED, "style_create: style layer %d is %dx%d, kernels are built for %dx%d", i,
                        style_layers[i].out_features, style_layers[i].in_features, want_s[i][0], want_s[i][1]);
    // hidden ReLU layers equalised by powers of two (mlp_pack.h, EqualisedNet); concat layer 4 (concat_features) and the rgb
    // head keep their rows: their outputs leave the operators.  Inputs are cat(h, latent[, x]): h occupies columns 0..255.
    EqualisedNet eqc, eqs;
    eqc.copy(concat_layers, 5), eqs.copy(style_layers, 8);
    for (int l = 0; l < 4; ++l) eqc.run(l, {{l + 1, 0}});
    for (int l = 0; l < 7; ++l) eqs.run(l, {{l + 1, 0}});
    concat_layers = eqc.lin.data(), style_layers = eqs.lin.data();
    const bool split = precision == TGTC_PREC_FP16X3;
    PackedNet pc = pack_layers(concat_specs(concat_layers), split);
    PackedNet ps = pack_layers(style_specs(style_layers), split);
    if (pc.n_frags != kConcatFrags || ps.n_frags != kStyleFrags || (int)pc.bias.size() != kConcatBiasFloats ||
        (int)ps.bias.size() != kStyleBiasFloats)
        return fail(TGTC_ERR_UNSUPPORTED, "style_create: internal layout mismatch");
    for (int i = 0; i < 8; ++i)
        if (ps.frag0[i] != style_frag0(i) || ps.bias0[i] != style_bias0(i))
            return fail(TGTC_ERR_UNSUPPORTED, "style_create: internal layout mismatch at style layer %d", i);
    int dev = 0, n_cu = 256;
    TGTC_HIP_CHECK(hipGetDevice(&dev));
    TGTC_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    tgtc_net* net = new tgtc_net();
    net->kind = 1, net->precision = precision;
    net->bias_bytes = kStylePairBiasBytes;
    net->stream_bytes = pc.stream.size() * sizeof(half_t);
    net->n_frags = pc.n_frags;
    net->stream2_off = net->bias_bytes + net->stream_bytes;
    net->stream2_bytes = ps.stream.size() * sizeof(half_t);
    net->n_frags2 = ps.n_frags;
    net->n_wg = n_cu;  // one persistent workgroup per CU (the LDS ring allows exactly one)
    net->stash_off = (net->stream2_off + net->stream2_bytes + kChunkBytes + 255) & ~(size_t)255;
    const size_t total = net->stash_off + (size_t)net->n_wg * kStashBytesPerWG;
    hipError_t e = hipMalloc((void**)&net->dev, total);
    if (e != hipSuccess) {
        delete net;
        return fail(TGTC_ERR_HIP, "style_create: hipMalloc(%zu): %s", total, hipGetErrorString(e));
    }
    std::vector<char> host(net->stash_off, 0);
    memcpy(host.data(), pc.bias.data(), pc.bias.size() * sizeof(float));
    memcpy(host.data() + kConcatBiasFloats * sizeof(float), ps.bias.data(), ps.bias.size() * sizeof(float));
    memcpy(host.data() + net->bias_bytes, pc.stream.data(), net->stream_bytes);
    memcpy(host.data() + net->stream2_off, ps.stream.data(), net->stream2_bytes);
    e = hipMemcpy(net->dev, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(net->dev);
        delete net;
        return fail(TGTC_ERR_HIP, "style_create: hipMemcpy: %s", hipGetErrorString(e));
    }
    *out = net;
    return TGTC_OK;
}

extern "C" int tgtc_concat_mlp_forward(const tgtc_net* style, const float* x, const float* latent, int64_t M,
                                       float* concat_features, void* stream) {
    TGTC_REQUIRE(style && style->kind == 1 && M >= 0, "concat_mlp_forward: bad argument");
    if (M == 0) return TGTC_OK;
    TGTC_REQUIRE(x && latent && concat_features, "concat_mlp_forward: null pointer");
    ConcatArgs a{style->dev, style->dev + style->bias_bytes, M, x, latent, concat_features};
    if (style->precision == TGTC_PREC_FP16) launch_concat<CfgFast>(a, as_stream(stream));
    else launch_concat<CfgExact>(a, as_stream(stream));
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_style_mlp_forward(const tgtc_net* style, const float* x, const float* concated,
                                      const float* latent, int64_t M, float* rgb, void* stream) {
    TGTC_REQUIRE(style && style->kind == 1 && M >= 0, "style_mlp_forward: bad argument");
    if (M == 0) return TGTC_OK;
    TGTC_REQUIRE(x && concated && latent && rgb, "style_mlp_forward: null pointer");
    StyleArgs a{style->dev, style->dev + style->stream2_off, M, x, concated, latent, rgb};
    if (style->precision == TGTC_PREC_FP16) launch_style<CfgFastNarrow>(a, as_stream(stream));
    else launch_style<CfgExactNarrow>(a, as_stream(stream));
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_styled_forward_rays(const tgtc_net* nerf, const tgtc_net* style, const double* rays_o,
                                        const double* rays_d, const float* ts, const float* z, int64_t R, int N,
                                        float* rgb, float* sigma, void* stream) {
    TGTC_REQUIRE(nerf && nerf->kind == 0 && style && style->kind == 1 && R >= 0 && N >= 1, "styled_forward_rays: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rays_o && rays_d && ts && z && rgb, "styled_forward_rays: null pointer");
    return styled_forward_rays_impl(nerf, style, rays_o, rays_d, ts, z, R, N, rgb, sigma, as_stream(stream));
}
#endif  // TGTC_TU_FP16_ONLY
