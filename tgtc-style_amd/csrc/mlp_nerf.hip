// Fused positional encoding + NeRF MLP (reference models.py:63-117 MLP_style inside :182-223 StyleNerf;
// D=8, W=256, skip at 4, view-dependent colour head).  One launch evaluates rgb / sigma for M samples;
// the [M,256] activations of the 11 dense layers never leave registers (mlp_core.h).
#include <cstdlib>

#include "mlp_core.h"
#include "mlp_pack.h"
#include "mlp_layouts.h"
#include "mlp_nerf_front.h"
#include "mlp_nerf_chain.h"
#include "mlp_nerf_mx.h"

namespace tgtc {

// ------------------------------------------------------------------------------------------------
// PARK geometry (MlpCfg::PARK): one wave per SIMD with the whole 512-register file.  The layer being produced is
// parked in AGPRs and copied into VGPRs at the layer boundary, where the previous input dies; every MFMA operand
// is an architectural VGPR.  Same arithmetic, same fragment stream and same results as the ping-pong path below.
template <class C>
struct ParkedAct {
    unsigned h[8][C::NCT][4];
    unsigned l[C::SPLIT ? 8 : 1][C::NCT][4];
};

template <class C, int KS>
__device__ __forceinline__ void unpark_act(const ParkedAct<C>& P, half8 (&Xh)[8][C::NCT], half8 (&Xl)[8][C::NCT]) {
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int c = 0; c < C::NCT; ++c) {
            Xh[k][c] = unpark4(P.h[k][c]);
            if constexpr (C::SPLIT) Xl[k][c] = unpark4(P.l[k][c]);
        }
}

template <class C, int IN_MODE, bool FULL, class WS>
__device__ __forceinline__ void nerf_layers_parked(WS& ws, lds_cptr bias_lane, const NerfArgs& a, const long long (&sidx)[C::NCT],
                                                   int g, half8 (&pe_h)[2][C::NCT], half8 (&pe_l)[2][C::NCT],
                                                   half8 (&de_h)[1][C::NCT], half8 (&de_l)[1][C::NCT]) {
    constexpr int NCT = C::NCT;
    constexpr bool SPLIT = C::SPLIT;
    using L = NerfLayout;
    ParkedAct<C> P;
    half8 Xh[8][NCT], Xl[8][NCT];
    auto to_P = [&](auto rt_, auto c_, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value, hf = decltype(h_)::value;
        unsigned ph, pl = 0;
        if constexpr (kAbl & 8) ph = pl = __builtin_bit_cast(unsigned, acc[2 * hf]);
        else pack_act<SPLIT, hf>(acc, ph, pl);
        P.h[rt / 2][c][(rt & 1) * 2 + hf] = park(ph);
        if constexpr (SPLIT) P.l[rt / 2][c][(rt & 1) * 2 + hf] = park(pl);
    };
    dense_layer<C, L::frag0(0), 2, 16, L::bias0(0)>(ws, bias_lane, pe_h, pe_l, to_P);
    // the point encoding is needed again by the skip layer only: park it meanwhile
    unsigned pe_ph[2][NCT][4], pe_pl[2][NCT][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            park4(pe_h[k][c], pe_ph[k][c]);
            if constexpr (SPLIT) park4(pe_l[k][c], pe_pl[k][c]);
        }
    unpark_act<C, 8>(P, Xh, Xl);
    dense_layer<C, L::frag0(1), 8, 16, L::bias0(1)>(ws, bias_lane, Xh, Xl, to_P);
    unpark_act<C, 8>(P, Xh, Xl);
    dense_layer<C, L::frag0(2), 8, 16, L::bias0(2)>(ws, bias_lane, Xh, Xl, to_P);
    unpark_act<C, 8>(P, Xh, Xl);
    dense_layer<C, L::frag0(3), 8, 16, L::bias0(3)>(ws, bias_lane, Xh, Xl, to_P);
    unpark_act<C, 8>(P, Xh, Xl);
    dense_layer<C, L::frag0(4), 8, 16, L::bias0(4)>(ws, bias_lane, Xh, Xl, to_P);
    {
        // skip layer: reference input is cat(pe, h) (models.py:98-99); k order here is [h | pe]
        half8 Bh[10][NCT], Bl[10][NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                Bh[k][c] = unpark4(P.h[k][c]);
                if constexpr (SPLIT) Bl[k][c] = unpark4(P.l[k][c]);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                Bh[8 + k][c] = unpark4(pe_ph[k][c]);
                if constexpr (SPLIT) Bl[8 + k][c] = unpark4(pe_pl[k][c]);
            }
        }
        dense_layer<C, L::frag0(5), 10, 16, L::bias0(5)>(ws, bias_lane, Bh, Bl, to_P);
    }
    unpark_act<C, 8>(P, Xh, Xl);
    dense_layer<C, L::frag0(6), 8, 16, L::bias0(6)>(ws, bias_lane, Xh, Xl, to_P);
    unpark_act<C, 8>(P, Xh, Xl);
    dense_layer<C, L::frag0(7), 8, 16, L::bias0(7)>(ws, bias_lane, Xh, Xl, to_P);
    unpark_act<C, 8>(P, Xh, Xl);
    dense_layer<C, L::frag0(8), 8, 1, L::bias0(8)>(ws, bias_lane, Xh, Xl, [&](auto, auto c_, auto h_, const float4v& acc) {
        constexpr int c = decltype(c_)::value;
        if constexpr (decltype(h_)::value == 0)
            if (g == 0 && a.sigma && sidx[c] < a.M) a.sigma[sidx[c]] = acc[0];
    });
    if constexpr (FULL) {
        dense_layer<C, L::frag0(9), 8, 16, L::bias0(9)>(ws, bias_lane, Xh, Xl, [&](auto rt_, auto c_, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value, hf = decltype(h_)::value;
            to_P(rt_, c_, h_, acc);
            if (a.remap && sidx[c] < a.M) {
                float* o = a.remap + sidx[c] * 256 + 16 * rt + 4 * g + 2 * hf;
                o[0] = relu(acc[2 * hf]), o[1] = relu(acc[2 * hf + 1]);
            }
        });
        {
            half8 Bh[9][NCT], Bl[9][NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                nerf_encode_dir_late<IN_MODE, SPLIT>(a, sidx[c], g, de_h[0][c], de_l[0][c]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    Bh[k][c] = unpark4(P.h[k][c]);
                    if constexpr (SPLIT) Bl[k][c] = unpark4(P.l[k][c]);
                }
                Bh[8][c] = de_h[0][c], Bl[8][c] = de_l[0][c];
            }
            dense_layer<C, L::frag0(10), 9, 8, L::bias0(10)>(ws, bias_lane, Bh, Bl, to_P);
        }
        half8 Zh[4][NCT], Zl[4][NCT];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                Zh[k][c] = unpark4(P.h[k][c]);
                if constexpr (SPLIT) Zl[k][c] = unpark4(P.l[k][c]);
            }
        dense_layer<C, L::frag0(11), 4, 1, L::bias0(11)>(ws, bias_lane, Zh, Zl, [&](auto, auto c_, auto h_, const float4v& acc) {
            constexpr int c = decltype(c_)::value, hf = decltype(h_)::value;
            if (g == 0 && a.rgb && sidx[c] < a.M) {
#pragma unroll
                for (int r = 2 * hf; r < (hf ? 3 : 2); ++r) a.rgb[sidx[c] * 3 + r] = 1.0f / (1.0f + expf(-acc[r]));
            }
        });
    }
}

template <class C, int IN_MODE, bool FULL>
__global__ void __launch_bounds__(C::NWAVES * 64, C::NWAVES * C::WG_PER_CU / 4) nerf_mlp_kernel(NerfArgs a) {
    constexpr int NCT = C::NCT;
    constexpr bool SPLIT = C::SPLIT;
    constexpr int NFRAG = FULL ? NerfLayout::kFragsFull : NerfLayout::kFragsSigma;
    using L = NerfLayout;

    // ALL LDS in one array (a second __shared__ object makes hipcc drain vmcnt before LDS reads).
    __shared__ __attribute__((aligned(16))) char smem[C::RING_BYTES + kNerfBiasBytes];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    const long long s_wave = (long long)blockIdx.x * C::SAMPLES_PER_WG + wave * C::SAMPLES_PER_WAVE;

    // ---- 1. inputs (ordinary loads first: once LDS-DMA is in flight hipcc drains vmcnt(0) for them)
    double pos[NCT][3], dir[NCT][3];
    long long sidx[NCT];
    nerf_load_samples<NCT, IN_MODE>(a, s_wave, n, pos, dir, sidx);
    half8 pe_h[2][NCT], pe_l[2][NCT], de_h[1][NCT], de_l[1][NCT];
    if constexpr (IN_MODE == IN_ENC) nerf_load_encoded<NCT, SPLIT, false>(a, sidx, g, pe_h, pe_l, de_h, de_l);

    // ---- 2. start the weight stream: bias table, then the first 8 chunks
    WeightStream<C, SingleStreamMap<NFRAG>> ws;
    const char* const streams[1] = {a.stream};
    ws.init(streams, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < kNerfBiasBytes / (C::NWAVES * 1024); ++j)
        lds_dma16(a.bias + (j * C::NWAVES + wave) * 1024 + lane * 16, smem + C::RING_BYTES + (j * C::NWAVES + wave) * 1024);
    ws.prologue();

    // ---- 3. positional encoding into B fragments (overlaps the prefetch latency)
    // (points only: the direction is encoded in front of the colour head, nerf_encode_dir_late -- eight to sixteen
    // registers that would otherwise sit idle through eleven layers)
    if constexpr (IN_MODE != IN_ENC) nerf_encode<NCT, SPLIT, false>(a, pos, dir, sidx, g, pe_h, pe_l, de_h, de_l);

    const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * g);
    ws.start();
    if constexpr (C::PARK) {
        nerf_layers_parked<C, IN_MODE, FULL>(ws, bias_lane, a, sidx, g, pe_h, pe_l, de_h, de_l);
        return;
    }

    // ---- 4. the twelve layers
    nerf_chain<C, FULL>(
        ws, bias_lane, pe_h, pe_l,
        [&](auto c_, half8& dh, half8& dl) {
            constexpr int c = decltype(c_)::value;
            nerf_encode_dir_late<IN_MODE, SPLIT>(a, sidx[c], g, de_h[0][c], de_l[0][c]);
            dh = de_h[0][c], dl = de_l[0][c];
        },
        [&](auto c_, float sigma) {
            constexpr int c = decltype(c_)::value;
            if (g == 0 && a.sigma && sidx[c] < a.M) a.sigma[sidx[c]] = sigma;
        },
        [&](auto rt_, auto c_, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, c = decltype(c_)::value, hf = decltype(h_)::value;
            if (a.remap && sidx[c] < a.M) {
                float* o = a.remap + sidx[c] * 256 + 16 * rt + 4 * g + 2 * hf;
                o[0] = relu(acc[2 * hf]), o[1] = relu(acc[2 * hf + 1]);
            }
        },
        [&](auto c_, auto h_, const float4v& acc) {
            constexpr int c = decltype(c_)::value, hf = decltype(h_)::value;
            if (g == 0 && a.rgb && sidx[c] < a.M) {
#pragma unroll
                for (int r = 2 * hf; r < (hf ? 3 : 2); ++r) a.rgb[sidx[c] * 3 + r] = 1.0f / (1.0f + expf(-acc[r]));
            }
        });
}

// ------------------------------------------------------------------------------------------------ host
#ifndef TGTC_TU_FP16_ONLY
std::vector<LayerSpec> nerf_specs(const tgtc_linear* l) {
    std::vector<LayerSpec> v;
    auto add = [&](int idx, std::vector<Seg> segs) {
        v.push_back(LayerSpec{l[idx].weight, l[idx].bias, l[idx].out_features, l[idx].in_features, std::move(segs)});
    };
    add(0, {{SEG_PE63, 0, 2}});
    for (int i = 1; i <= 4; ++i) add(i, {{SEG_ACT, 0, 8}});
    add(5, {{SEG_ACT, 63, 8}, {SEG_PE63, 0, 2}});  // reference column order: [pe(63) | h(256)]
    add(6, {{SEG_ACT, 0, 8}});
    add(7, {{SEG_ACT, 0, 8}});
    add(8, {{SEG_ACT, 0, 8}});                       // sigma_layer
    add(9, {{SEG_ACT, 0, 8}});                       // base_remap_layer
    add(10, {{SEG_ACT, 0, 8}, {SEG_PE27, 256, 1}});  // rgb_layers.0 on [remap(256) | dirs(27)]
    add(11, {{SEG_ACT, 0, 4}});                      // rgb_layers.1
    return v;
}

#endif

template <class C, int IN_MODE, bool FULL>
int launch_nerf(const NerfArgs& a, hipStream_t st) {
    const long long nwg = (a.M + C::SAMPLES_PER_WG - 1) / C::SAMPLES_PER_WG;
    if (a.M >= 0x7fffffffLL) return fail(TGTC_ERR_UNSUPPORTED, "nerf: too many samples in one launch (%lld)", a.M);
    nerf_mlp_kernel<C, IN_MODE, FULL><<<(unsigned)nwg, C::NWAVES * 64, 0, st>>>(a);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

// Geometry chosen by measurement (tools/bench_variants.sh, profiles/r1_kernel_variants.md): 8 waves per
// workgroup = 2 per SIMD, every MFMA operand in arch VGPRs (<= 256 registers per wave).  MFMAs whose B
// operand sits in the accumulator half of the register file issue ~25 % slower (tools/microbench/mfma_regs),
// which is what the 4-wave / 4-column-tile geometry (470-510 registers) ran into.
using CfgFast = MlpCfg<8, 2, false, 4>;
using CfgExact = MlpCfg<8, 1, true, 4>;

#define TGTC_NERF_FP16_INSTANCES(PREFIX)                                           \
    PREFIX template int launch_nerf<CfgFast, IN_RAYS, false>(const NerfArgs&, hipStream_t); \
    PREFIX template int launch_nerf<CfgFast, IN_RAYS, true>(const NerfArgs&, hipStream_t);  \
    PREFIX template int launch_nerf<CfgFast, IN_PTS, true>(const NerfArgs&, hipStream_t);   \
    PREFIX template int launch_nerf<CfgFast, IN_ENC, true>(const NerfArgs&, hipStream_t);
#ifdef TGTC_TU_FP16_ONLY
TGTC_NERF_FP16_INSTANCES()
}  // namespace tgtc
#else
#ifndef TGTC_DEV_VARIANT   // development builds compile their one configuration here and leave mlp_nerf_fp16.o out
TGTC_NERF_FP16_INSTANCES(extern)
#endif

// 1 (shipped): sigma-only fp16x3 launches over rays (the coarse pass of a render) go to the two-tile persistent kernel (mlp_nerf_x3s.hip)
#ifndef TGTC_X3S
#define TGTC_X3S 1
#endif

template <int IN_MODE, bool FULL>
static int dispatch_nerf(const tgtc_net* net, NerfArgs& a, hipStream_t st) {
    a.bias = net->dev;
    a.stream = net->dev + net->bias_bytes;
#ifdef TGTC_DEV_VARIANT   // development builds: only the hot-path kernels of one experimental configuration
    if constexpr (IN_MODE == IN_RAYS) {
        if (net->precision == (TGTC_DEV_VARIANT::SPLIT ? TGTC_PREC_FP16X3 : TGTC_PREC_FP16))
            return launch_nerf<TGTC_DEV_VARIANT, IN_MODE, FULL>(a, st);
    }
    return fail(TGTC_ERR_UNSUPPORTED, "development build: kernel not compiled");
#else
    if (net->precision == TGTC_PREC_FP16_FP6) {
        if (a.M >= 0x7fffffffLL) return fail(TGTC_ERR_UNSUPPORTED, "nerf: too many samples in one launch (%lld)", a.M);
        return nerf_mx_launch(IN_MODE, FULL, a, st);
    }
    if (net->precision == TGTC_PREC_FP16) return launch_nerf<CfgFast, IN_MODE, FULL>(a, st);
    if constexpr (TGTC_X3S && IN_MODE == IN_RAYS && !FULL) {   // the coarse pass: sigma only, no encodings out
        if (a.sigma && !a.out_pts_enc && !a.out_dirs_enc) {
            if (a.M >= 0x7fffffffLL) return fail(TGTC_ERR_UNSUPPORTED, "nerf: too many samples in one launch (%lld)", a.M);
            return nerf_x3s_launch(a, st);
        }
    }
    return launch_nerf<CfgExact, IN_MODE, FULL>(a, st);
#endif
}

int nerf_forward_rays_impl(const tgtc_net* net, const double* rays_o, const double* rays_d, const float* ts, int64_t R,
                           int N, float* rgb, float* sigma, hipStream_t st) {
    NerfArgs a{};
    a.M = R * (int64_t)N, a.N = N, a.rays_o = rays_o, a.rays_d = rays_d, a.ts = ts, a.rgb = rgb, a.sigma = sigma;
    return rgb ? dispatch_nerf<IN_RAYS, true>(net, a, st) : dispatch_nerf<IN_RAYS, false>(net, a, st);
}

}  // namespace tgtc

using namespace tgtc;

extern "C" int tgtc_nerf_create(const tgtc_linear* layers, int n_layers, int precision, tgtc_net** out) {
    TGTC_REQUIRE(layers && out, "nerf_create: null argument");
    TGTC_REQUIRE(precision == TGTC_PREC_FP16 || precision == TGTC_PREC_FP16X3 || precision == TGTC_PREC_FP16_FP6,
                 "nerf_create: unknown precision %d", precision);
    static const int want[12][2] = {{256, 63},  {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 319},
                                    {256, 256}, {256, 256}, {1, 256},   {256, 256}, {128, 283}, {3, 128}};
    if (n_layers != 12) return fail(TGTC_ERR_UNSUPPORTED, "nerf_create: expected the 12 linears of MLP_style (D=8), got %d", n_layers);
    for (int i = 0; i < 12; ++i) {
        TGTC_REQUIRE(layers[i].weight && layers[i].bias, "nerf_create: layer %d has a null pointer", i);
        if (layers[i].out_features != want[i][0] || layers[i].in_features != want[i][1])
            return fail(TGTC_ERR_UNSUPPORTED, "nerf_create: layer %d is %dx%d, kernels are built for %dx%d (D=8, W=256, PE 10/4, viewdirs)",
                        i, layers[i].out_features, layers[i].in_features, want[i][0], want[i][1]);
    }
    std::vector<char> bias_region, stream;
    int n_frags = 0;
    // per-feature scales of the ReLU trunk balanced by powers of two first (mlp_pack.h, EqualisedNet): the same function,
    // numbers that every precision mode represents well.  base_remap's rows stay: it is an output of the operator.
    EqualisedNet eq;
    eq.copy(layers, 12);
    for (int l = 0; l < 7; ++l) eq.run(l, {{l + 1, l + 1 == 5 ? 63 : 0}});   // layer 5 reads cat(pe(63), h) (models.py:98-99)
    eq.run(7, {{8, 0}, {9, 0}});                                              // sigma_layer and base_remap_layer read h
    eq.run(10, {{11, 0}});                                                    // rgb_layers.0 -> rgb_layers.1
    layers = eq.lin.data();
    if (precision == TGTC_PREC_FP16_FP6) {
        const int rc = nerf_mx_pack(layers, bias_region, stream);
        if (rc != TGTC_OK) return rc;
    } else {
        PackedNet p = pack_layers(nerf_specs(layers), precision == TGTC_PREC_FP16X3);
        if (p.n_frags != NerfLayout::kFragsFull || (int)p.bias.size() != NerfLayout::kBiasFloats)
            return fail(TGTC_ERR_UNSUPPORTED, "nerf_create: internal layout mismatch (%d frags, %zu bias)", p.n_frags, p.bias.size());
        for (int i = 0; i < 12; ++i)
            if (p.frag0[i] != NerfLayout::frag0(i) || p.bias0[i] != NerfLayout::bias0(i))
                return fail(TGTC_ERR_UNSUPPORTED, "nerf_create: internal layout mismatch at layer %d", i);
        bias_region.assign(kNerfBiasBytes, 0);
        std::memcpy(bias_region.data(), p.bias.data(), p.bias.size() * sizeof(float));
        stream.assign(reinterpret_cast<const char*>(p.stream.data()),
                      reinterpret_cast<const char*>(p.stream.data()) + p.stream.size() * sizeof(half_t));
        n_frags = p.n_frags;
    }
    tgtc_net* net = new tgtc_net();
    net->kind = 0, net->precision = precision;
    net->bias_bytes = kNerfBiasBytes;
    net->stream_bytes = stream.size();
    net->n_frags = n_frags;
    // + one ring of slack: the fused ray kernel rounds a pass up to a whole number of rings and fetches (never reads)
    // the chunks behind the stream's end (mlp_core.h, WeightStream PERSIST)
    size_t total = net->bias_bytes + net->stream_bytes + kRingBytes;
    hipError_t e = hipMalloc((void**)&net->dev, total);
    if (e != hipSuccess) {
        delete net;
        return fail(TGTC_ERR_HIP, "nerf_create: hipMalloc(%zu): %s", total, hipGetErrorString(e));
    }
    std::vector<char> host(total, 0);
    std::memcpy(host.data(), bias_region.data(), bias_region.size());
    std::memcpy(host.data() + net->bias_bytes, stream.data(), net->stream_bytes);
    e = hipMemcpy(net->dev, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(net->dev);
        delete net;
        return fail(TGTC_ERR_HIP, "nerf_create: hipMemcpy: %s", hipGetErrorString(e));
    }
    *out = net;
    return TGTC_OK;
}

extern "C" int tgtc_net_destroy(tgtc_net* net) {
    if (!net) return TGTC_OK;
    hipError_t e = net->dev ? hipFree(net->dev) : hipSuccess;
    delete net;
    if (e != hipSuccess) return fail(TGTC_ERR_HIP, "net_destroy: hipFree: %s", hipGetErrorString(e));
    return TGTC_OK;
}

extern "C" int tgtc_net_precision(const tgtc_net* net) { return net ? net->precision : TGTC_ERR_ARG; }

extern "C" int tgtc_nerf_forward(const tgtc_net* net, const double* pts, const double* dirs, int64_t M, float* rgb,
                                 float* sigma, float* base_remap, float* pts_enc, float* dirs_enc, void* stream) {
    TGTC_REQUIRE(net && net->kind == 0 && M >= 0, "nerf_forward: bad argument");
    if (M == 0) return TGTC_OK;  // empty tensors carry null pointers
    TGTC_REQUIRE(pts && dirs, "nerf_forward: null input");
    NerfArgs a{};
    a.M = M, a.pts = pts, a.dirs = dirs, a.rgb = rgb, a.sigma = sigma, a.remap = base_remap;
    a.out_pts_enc = pts_enc, a.out_dirs_enc = dirs_enc;
    return dispatch_nerf<IN_PTS, true>(net, a, as_stream(stream));
}

extern "C" int tgtc_nerf_mlp_forward(const tgtc_net* net, const float* pts_enc, const float* dirs_enc, int64_t M,
                                     float* rgb, float* sigma, float* base_remap, void* stream) {
    TGTC_REQUIRE(net && net->kind == 0 && M >= 0, "nerf_mlp_forward: bad argument");
    if (M == 0) return TGTC_OK;
    TGTC_REQUIRE(pts_enc && dirs_enc, "nerf_mlp_forward: null input");
    NerfArgs a{};
    a.M = M, a.pts_enc = pts_enc, a.dirs_enc = dirs_enc, a.rgb = rgb, a.sigma = sigma, a.remap = base_remap;
    return dispatch_nerf<IN_ENC, true>(net, a, as_stream(stream));
}

extern "C" int tgtc_nerf_forward_rays(const tgtc_net* net, const double* rays_o, const double* rays_d, const float* ts,
                                      int64_t R, int N, float* rgb, float* sigma, void* stream) {
    TGTC_REQUIRE(net && net->kind == 0 && R >= 0 && N >= 1, "nerf_forward_rays: bad argument");
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rays_o && rays_d && ts && (rgb || sigma), "nerf_forward_rays: null input");
    return nerf_forward_rays_impl(net, rays_o, rays_d, ts, R, N, rgb, sigma, as_stream(stream));
}
#endif  // TGTC_TU_FP16_ONLY
