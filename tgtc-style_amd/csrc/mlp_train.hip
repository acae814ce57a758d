// Fused TRAINING path of the NeRF MLP (reference train_tgtcs.py:218-309 `Origin_train`
// backpropagates through MLP_style, models.py:95-117): forward with activation stash, input-gradient chain and weight
// gradients as three launches per network, every product on fp16x3 MFMAs (fp32-equivalent, like the render path).
//
//   tgtc_trainer_forward    gather-pack of the CURRENT fp32 parameters into the fragment-ordered fp16 hi/lo stream (one small
//                           kernel: the weights change every optimiser step), then the fused PE + 12-layer kernel of the
//                           render path, which additionally leaves every layer's activations (fp16 hi/lo, train_layouts.h) and
//                           their ReLU gates in the workspace;
//   tgtc_trainer_backward   (1) the same chain run BACKWARDS on the transposed weights: d rgb / d sigma -> gated
//                           pre-activation gradients of every layer, held in registers as fp16 hi/lo with a per-tile power-of-two
//                           scale, written once as fp32;  (2) dW = dZ^T H as output-stationary split-K MFMA tiles over the
//                           stashed rows (LDS transposes by ds_read_b64_tr_b16), db as column sums, accumulated with atomics.
//
// The unfused form of the same arithmetic (one GEMM launch per product, autograd_ops.py) stays as the reference
// implementation the gradient tests compare this path with.
#include "../../include/tgtc_train.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "mlp_core.h"
#include "mlp_layouts.h"
#include "mlp_nerf_front.h"
#include "mlp_pack.h"
#include "train_layouts.h"

namespace tgtc {
namespace train {

using CfgT = MlpCfg<8, 1, true, 4>;   // the fp16x3 geometry of the render kernels: 8 waves x 16 samples, 128 per workgroup
constexpr int kTileSamples = CfgT::SAMPLES_PER_WG;

// ------------------------------------------------------------------------------------------------ gather-pack
// One entry per fp16 slot of a fragment stream: which parameter tensor and which element of it (or nothing) goes there.
// The maps are built once per trainer on the host by walking the packer's loops (mlp_pack.h pack_layers), so the device
// stream is what tgtc_nerf_create would pack from the same weights WITHOUT its power-of-two equalisation (EqualisedNet): the
// gradients are those of the caller's tensors, row for row.
struct PackSrc {
    int pid;   // index into the 24 parameter pointers (2 * layer: weight, 2 * layer + 1: bias), -1: zero
    int off;
};
struct ParamPtrs {
    const float* p[24];
};

__global__ void __launch_bounds__(256) gather_pack_kernel(ParamPtrs params, const PackSrc* __restrict__ map, int n_frag_slots,
                                                          half_t* __restrict__ stream, const PackSrc* __restrict__ bias_map,
                                                          int n_bias, float* __restrict__ bias) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_frag_slots) {   // slot i = fragment f, lane, element j:  hi at f*1024 + lane*8 + j, lo 512 halves behind
        const PackSrc s = map[i];
        const float w = s.pid >= 0 ? params.p[s.pid][s.off] : 0.0f;
        const half_t hi = (half_t)w;
        const int f = i >> 9, r = i & 511;
        stream[(size_t)f * 1024 + r] = hi;
        stream[(size_t)f * 1024 + 512 + r] = (half_t)(w - (float)hi);
    }
    if (i < n_bias) {
        const PackSrc s = bias_map[i];
        bias[i] = s.pid >= 0 ? params.p[s.pid][s.off] : 0.0f;
    }
}

// a layer as the map builder sees it: `segs` as in LayerSpec; element (row, col) of the (possibly transposed) matrix
struct MapLayer {
    int pid;                  // weight tensor
    int out, in;              // shape of the stored nn.Linear weight [out][in]
    int rows;                 // output features of THIS product (rows of the packed matrix)
    std::vector<Seg> segs;
    bool transposed;          // packed(row, col) = W[col][row + t_col0]   (input-gradient chain)
    int t_col0;
    int head_row;             // segs of kind SEG_VEC32: natural column c maps to weight row (c - head_row) of `head_pid`
    int head_pid, head_out, head_in;
};

static void build_stream_map(const std::vector<MapLayer>& layers, std::vector<PackSrc>& map) {
    map.clear();
    for (const MapLayer& L : layers) {
        const int RT = (L.rows + 15) / 16;
        for (int rt = 0; rt < RT; ++rt)
            for (const Seg& s : L.segs)
                for (int k = 0; k < s.ksteps; ++k) {
                    const size_t base = map.size();
                    map.resize(base + 512, PackSrc{-1, 0});
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane & 15, g = lane >> 4, row = 16 * rt + m;
                        for (int j = 0; j < 8; ++j) {
                            const int col = seg_col(s, k, g, j);
                            PackSrc src{-1, 0};
                            if (row < L.rows && col >= 0) {
                                if (!L.transposed) {
                                    if (col < L.in) src = PackSrc{L.pid, row * L.in + col};
                                } else if (s.kind == SEG_VEC32) {      // head k-step: natural column c = col - s.col0
                                    const int c = col - s.col0, hr = c - L.head_row;
                                    if (hr >= 0 && hr < L.head_out && row < L.head_in) src = PackSrc{L.head_pid, hr * L.head_in + row};
                                } else {
                                    const int c = col - s.col0;           // output feature of the forward layer
                                    if (c >= 0 && c < L.out && L.t_col0 + row < L.in) src = PackSrc{L.pid, c * L.in + L.t_col0 + row};
                                }
                            }
                            map[base + lane * 8 + j] = src;
                        }
                    }
                }
    }
    const size_t chunk_slots = kChunkBytes / 4;   // a chunk holds FPC fragments of 512 slots (hi + lo = 2 KiB each)
    map.resize((map.size() + chunk_slots - 1) / chunk_slots * chunk_slots, PackSrc{-1, 0});
}

// ------------------------------------------------------------------------------------------------ forward + stash
struct FwdArgs {
    const char* bias;      // padded fp32 bias table (kNerfBiasBytes)
    const char* stream;    // fragment stream
    long long M;
    const double* pts;     // [M,3]
    const double* dirs;    // [M,3]
    half_t* h_hi;          // segments of [M_pad][width] (train_layouts.h)
    half_t* h_lo;
    unsigned long long* gates;   // [kGateLayers][M_pad / 16][64]
    long long n_tiles;     // M_pad / 16
    float* rgb;            // [M,3]
    float* sigma;          // [M]
};

// (inline asm: __builtin_elementwise_min on a ushort2 is not selected as v_pk_min_u16 -- hipcc 7.2 expands it into SDWA
// compares and selects, ~20 instructions per register: the gate words were 57 % of this kernel's VALU instructions)
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned ones) {
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(ones));
    return r;
}
// ReLU gate bits of one k-step of non-negative fp16 activations: bit (j >> 1) + 4 * (j & 1) <- element j is positive
__device__ __forceinline__ unsigned gate_byte(half8 v) {
    const u4 w = __builtin_bit_cast(u4, v);
    const unsigned ones = 0x00010001u;
    const unsigned t0 = pk_min_u16(w[0], ones), t1 = pk_min_u16(w[1], ones);
    const unsigned t2 = pk_min_u16(w[2], ones), t3 = pk_min_u16(w[3], ones);
    const unsigned b = t0 | (t1 << 1) | (t2 << 2) | (t3 << 3);   // bits 0..3: elements 0,2,4,6; bits 16..19: 1,3,5,7
    return (b | (b >> 12)) & 0xffu;
}
template <int KS>
__device__ __forceinline__ unsigned long long gate_word(const half8 (&h)[KS][1]) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const unsigned b = gate_byte(h[ks][0]);
        if (ks < 4) lo |= b << (8 * ks);
        else hi |= b << (8 * (ks - 4));
    }
    return ((unsigned long long)hi << 32) | lo;
}

// this lane's eight values of every k-step of a finished KS-k-step feature set -> its sample's row of the segment at col0
template <int KS>
__device__ __forceinline__ void stash_set(half_t* hi, half_t* lo, long long m_pad, long long row, int col0, int g, const half8 (&h)[KS][1],
                                          const half8 (&l)[KS][1]) {
    const size_t at = seg_at(col0, 32 * KS, m_pad, row) + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        *reinterpret_cast<half8*>(hi + at + 32 * ks) = h[ks][0];
        *reinterpret_cast<half8*>(lo + at + 32 * ks) = l[ks][0];
    }
}

__global__ void __launch_bounds__(CfgT::NWAVES * 64, 2) train_forward_kernel(FwdArgs a) {
    using C = CfgT;
    using L = NerfLayout;
    __shared__ __attribute__((aligned(16))) char smem[C::RING_BYTES + kNerfBiasBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    const long long tile = (long long)blockIdx.x * C::NWAVES + wave;
    const long long s_wave = tile * 16;

    // ---- inputs (ordinary loads before any LDS-DMA, mlp_nerf_front.h)
    NerfArgs na{};
    na.M = a.M, na.pts = a.pts, na.dirs = a.dirs;
    double pos[1][3], dir[1][3];
    long long sidx[1];
    nerf_load_samples<1, IN_PTS>(na, s_wave, n, pos, dir, sidx);
    half8 pe_h[2][1], pe_l[2][1], de_h[1][1], de_l[1][1];

    WeightStream<C, SingleStreamMap<L::kFragsFull>> ws;
    const char* const streams[1] = {a.stream};
    ws.init(streams, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < kNerfBiasBytes / (C::NWAVES * 1024); ++j)
        lds_dma16(a.bias + (j * C::NWAVES + wave) * 1024 + lane * 16, smem + C::RING_BYTES + (j * C::NWAVES + wave) * 1024);
    ws.prologue();
    nerf_encode<1, true, true>(na, pos, dir, sidx, g, pe_h, pe_l, de_h, de_l);   // point AND direction encodings up front

    // this lane's sample row of the stash (rows up to M_pad exist: tail lanes store their duplicate of the last sample,
    // the backward gives those rows zero gradients)
    const long long m_pad = a.n_tiles * 16, row = s_wave + n;
    unsigned long long* const gate_lane = a.gates + (size_t)tile * 64 + lane;
    auto gate_out = [&](int layer, unsigned long long w) { gate_lane[(size_t)layer * a.n_tiles * 64] = w; };
    stash_set<2>(a.h_hi, a.h_lo, m_pad, row, H_PE, g, pe_h, pe_l);
    stash_set<1>(a.h_hi, a.h_lo, m_pad, row, H_DIR, g, de_h, de_l);

    const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * g);
    ws.start();

    half8 Xh[8][1], Xl[8][1], Yh[8][1], Yl[8][1];
    auto to_Y = [&](auto rt_, auto, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Yh[rt / 2][0], Yl[rt / 2][0]);
    };
    auto to_X = [&](auto rt_, auto, auto h_, const float4v& acc) {
        constexpr int rt = decltype(rt_)::value;
        store_act<C, rt, decltype(h_)::value>(acc, Xh[rt / 2][0], Xl[rt / 2][0]);
    };
    // a finished 256-wide layer: its fragments ARE the stash rows (fragment order), its gates one 64-bit word per lane
    auto keep = [&](int layer, int col0, const half8 (&h)[8][1], const half8 (&l)[8][1]) {
        stash_set<8>(a.h_hi, a.h_lo, m_pad, row, col0, g, h, l);
        gate_out(layer, gate_word<8>(h));
    };

    dense_layer<C, L::frag0(0), 2, 16, L::bias0(0)>(ws, bias_lane, pe_h, pe_l, to_Y);
    keep(0, h_layer(0), Yh, Yl);
    dense_layer<C, L::frag0(1), 8, 16, L::bias0(1)>(ws, bias_lane, Yh, Yl, to_X);
    keep(1, h_layer(1), Xh, Xl);
    dense_layer<C, L::frag0(2), 8, 16, L::bias0(2)>(ws, bias_lane, Xh, Xl, to_Y);
    keep(2, h_layer(2), Yh, Yl);
    dense_layer<C, L::frag0(3), 8, 16, L::bias0(3)>(ws, bias_lane, Yh, Yl, to_X);
    keep(3, h_layer(3), Xh, Xl);
    dense_layer<C, L::frag0(4), 8, 16, L::bias0(4)>(ws, bias_lane, Xh, Xl, to_Y);
    keep(4, h_layer(4), Yh, Yl);
    {
        half8 Bh[10][1], Bl[10][1];   // skip layer: reference input cat(pe, h) (models.py:98-99), k order here [h | pe]
#pragma unroll
        for (int k = 0; k < 8; ++k) Bh[k][0] = Yh[k][0], Bl[k][0] = Yl[k][0];
        Bh[8][0] = pe_h[0][0], Bh[9][0] = pe_h[1][0], Bl[8][0] = pe_l[0][0], Bl[9][0] = pe_l[1][0];
        dense_layer<C, L::frag0(5), 10, 16, L::bias0(5)>(ws, bias_lane, Bh, Bl, to_X);
    }
    keep(5, h_layer(5), Xh, Xl);
    dense_layer<C, L::frag0(6), 8, 16, L::bias0(6)>(ws, bias_lane, Xh, Xl, to_Y);
    keep(6, h_layer(6), Yh, Yl);
    dense_layer<C, L::frag0(7), 8, 16, L::bias0(7)>(ws, bias_lane, Yh, Yl, to_X);
    keep(7, h_layer(7), Xh, Xl);
    dense_layer<C, L::frag0(8), 8, 1, L::bias0(8)>(ws, bias_lane, Xh, Xl, [&](auto, auto, auto h_, const float4v& acc) {
        if constexpr (decltype(h_)::value == 0)
            if (g == 0 && sidx[0] < a.M) a.sigma[sidx[0]] = acc[0];                    // models.py:103
    });
    dense_layer<C, L::frag0(9), 8, 16, L::bias0(9)>(ws, bias_lane, Xh, Xl, to_Y);      // base_remap (models.py:106)
    keep(8, H_REMAP, Yh, Yl);
    half8 Zh[4][1], Zl[4][1];
    {
        half8 Bh[9][1], Bl[9][1];
#pragma unroll
        for (int k = 0; k < 8; ++k) Bh[k][0] = Yh[k][0], Bl[k][0] = Yl[k][0];
        Bh[8][0] = de_h[0][0], Bl[8][0] = de_l[0][0];
        dense_layer<C, L::frag0(10), 9, 8, L::bias0(10)>(ws, bias_lane, Bh, Bl, [&](auto rt_, auto, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value;
            store_act<C, rt, decltype(h_)::value>(acc, Zh[rt / 2][0], Zl[rt / 2][0]);
        });
    }
    stash_set<4>(a.h_hi, a.h_lo, m_pad, row, H_F, g, Zh, Zl);
    gate_out(9, gate_word<4>(Zh));
    dense_layer<C, L::frag0(11), 4, 1, L::bias0(11)>(ws, bias_lane, Zh, Zl, [&](auto, auto, auto h_, const float4v& acc) {
        constexpr int hf = decltype(h_)::value;
        if (g == 0 && sidx[0] < a.M) {
#pragma unroll
            for (int r = 2 * hf; r < (hf ? 3 : 2); ++r) a.rgb[sidx[0] * 3 + r] = 1.0f / (1.0f + expf(-acc[r]));   // models.py:111
        }
    });
}

}  // namespace train
}  // namespace tgtc

// ------------------------------------------------------------------------------------------------ handle
struct tgtc_trainer {
    char* dev = nullptr;              // one allocation
    size_t fwd_map_off = 0, fwd_bias_map_off = 0, bwd_map_off = 0, fwd_stream_off = 0, bwd_stream_off = 0, unperm_off = 0, maxima_off = 0;
    int fwd_slots = 0, bwd_slots = 0;   // fragment slots (fragments * 512) incl. chunk padding
    size_t fwd_stream_bytes = 0, bwd_stream_bytes = 0;
};

namespace tgtc {
namespace train {

struct WsLayout {
    long long m_pad, n_tiles;
    size_t h_hi, h_lo, dz, gates, total;
};
static WsLayout ws_layout(long long M) {
    WsLayout w;
    w.m_pad = (M + kTileSamples - 1) / kTileSamples * kTileSamples;
    w.n_tiles = w.m_pad / 16;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    };
    w.h_hi = take((size_t)w.m_pad * H_COLS * sizeof(half_t));
    w.h_lo = take((size_t)w.m_pad * H_COLS * sizeof(half_t));
    w.dz = take((size_t)w.m_pad * Z_COLS * sizeof(float));
    w.gates = take((size_t)kGateLayers * w.n_tiles * 64 * sizeof(unsigned long long));
    w.total = off;
    return w;
}

static std::vector<MapLayer> forward_layers() {
    // nerf_specs (mlp_nerf.hip) as map layers: the 12 linears in the order of MLP_style.layers
    static const int shape[12][2] = {{256, 63}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 319}, {256, 256}, {256, 256},
                                     {1, 256},  {256, 256}, {128, 283}, {3, 128}};
    std::vector<MapLayer> v;
    auto add = [&](int idx, std::vector<Seg> segs) {
        v.push_back(MapLayer{2 * idx, shape[idx][0], shape[idx][1], shape[idx][0], std::move(segs), false, 0, 0, -1, 0, 0});
    };
    add(0, {{SEG_PE63, 0, 2}});
    for (int i = 1; i <= 4; ++i) add(i, {{SEG_ACT, 0, 8}});
    add(5, {{SEG_ACT, 63, 8}, {SEG_PE63, 0, 2}});
    add(6, {{SEG_ACT, 0, 8}});
    add(7, {{SEG_ACT, 0, 8}});
    add(8, {{SEG_ACT, 0, 8}});
    add(9, {{SEG_ACT, 0, 8}});
    add(10, {{SEG_ACT, 0, 8}, {SEG_PE27, 256, 1}});
    add(11, {{SEG_ACT, 0, 4}});
    return v;
}

static std::vector<MapLayer> dgrad_layers() {
    // train_layouts.h D0..D9; weight tensors: pid = 2 * layer
    std::vector<MapLayer> v;
    // D0: rgb_layers.1^T: out 128 (f); one head k-step, natural columns 1..3 = rows 0..2 of rgb_layers.1 [3][128]
    v.push_back(MapLayer{-1, 0, 0, 128, {{SEG_VEC32, 0, 1}}, true, 0, 1, 22, 3, 128});
    // D1: rgb_layers.0^T (activation part): out 256 (remap), k = dz_f (4 k-steps); rgb_layers.0 is [128][283], columns 0..255
    v.push_back(MapLayer{20, 128, 283, 256, {{SEG_ACT, 0, 4}}, true, 0, 0, -1, 0, 0});
    // D2: base_remap^T | sigma^T: out 256 (h7), k = [dz_remap (8) | heads (1): natural column 0 = row 0 of sigma_layer [1][256]]
    v.push_back(MapLayer{18, 256, 256, 256, {{SEG_ACT, 0, 8}, {SEG_VEC32, 256, 1}}, true, 0, 0, 16, 1, 256});
    // D3..D9: base_layers[7..1]^T; layer 5 reads cat(pe(63), h): its activation columns start at 63
    for (int l = 7; l >= 1; --l) v.push_back(MapLayer{2 * l, 256, l == 5 ? 319 : 256, 256, {{SEG_ACT, 0, 8}}, true, l == 5 ? 63 : 0, 0, -1, 0, 0});
    return v;
}

}  // namespace train
}  // namespace tgtc

// ------------------------------------------------------------------------------------------------ input-gradient chain
namespace tgtc {
namespace train {

struct BwdArgs {
    const char* zero_bias;   // kNerfBiasBytes of zeros (the transposed products carry no bias)
    const char* stream;      // dgrad fragment stream (train_layouts.h D0..D9)
    long long M;
    const float* rgb;        // [M,3] forward output (sigmoid already applied)
    const float* d_rgb;      // [M,3] dL/d rgb
    const float* d_sigma;    // [M]   dL/d sigma
    const unsigned long long* gates;
    long long n_tiles;
    float* dz;               // segments of [M_pad][width] (train_layouts.h)
    unsigned* maxima;        // [12] bit patterns of max |dz| per dZ segment (dz0..7, remap, f, colour head, sigma head); non-negative floats order as ints
    unsigned* status;        // [1]  set to 1 if a scaled gradient left the fp16 range
};

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    return m;
}
// power-of-two scale exponent that brings a tile maximum to ~2^8 (fp16 hi/lo operands: 2^7 of headroom, 2^22 below)
__device__ __forceinline__ int scale_exp(float tile_max, int keep) {
    const unsigned b = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, tile_max));
    // (bounded: a tile whose largest gradient is below 2^-92 -- samples behind an opaque surface reach fp32 denormals -- is
    // scaled by 2^100 and no further.  Unbounded, a maximum in [2^-120, 2^-119) asked for 2^128 = inf, and the tile's
    // zero entries times inf put NaNs into every gradient: round 4 found the guard firing in ordinary Origin_train steps)
    return b == 0 ? keep : max(-100, min(100, 8 - ((int)(b >> 23) - 127)));
}
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(max(-126, min(127, e)) + 127) << 23); }

// signed hi/lo split of a pair (no ReLU): hi = fp16(v), lo = fp16(v - hi) (split_pair without the integer max)
__device__ __forceinline__ void split_pair_signed(float v0, float v1, unsigned& hpk, unsigned& lpk) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    hpk = __builtin_bit_cast(unsigned, __builtin_convertvector((float2v{v0, v1}), half2v));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lpk) : "v"(v0), "v"(hpk));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lpk) : "v"(v1), "v"(hpk));
}

// Epilogue of one accumulator half (rows 4g + 2*HALF, +1 of row tile RT) of a dgrad layer whose inputs carried the scale
// 2^e_in: ReLU gate of the layer below, true value to the dZ stash (16 bytes per row tile), tile maximum, and the operand of
// the next layer at scale 2^e_out.  `pend` carries the first half's pair until the second completes the 16-byte store.
template <int RT, int HALF, bool OPERAND>
__device__ __forceinline__ void dgrad_epi(const float4v& acc, unsigned long long gate, float s_true, float s_op, float& m, float (&pend)[2],
                                          float* seg_row, int g, half8& oh, half8& ol) {
    constexpr int ks = RT / 2, e0 = (RT & 1) * 4 + 2 * HALF;
    constexpr int bit0 = 8 * (ks & 3) + (e0 >> 1), bit1 = bit0 + 4;
    const int word = (int)(ks < 4 ? (unsigned)gate : (unsigned)(gate >> 32));
    const int k0 = __builtin_amdgcn_sbfe(word, bit0, 1), k1 = __builtin_amdgcn_sbfe(word, bit1, 1);   // 0 / -1
    // (copies first: __builtin_bit_cast applied directly to a vector ELEMENT reads element 0 whatever the index -- hipcc 7.2)
    const float a0 = acc[2 * HALF], a1 = acc[2 * HALF + 1];
    const float v0 = __builtin_bit_cast(float, __builtin_bit_cast(int, a0) & k0);
    const float v1 = __builtin_bit_cast(float, __builtin_bit_cast(int, a1) & k1);
    const float t0 = v0 * s_true, t1 = v1 * s_true;
    m = fmaxf(m, fmaxf(fabsf(t0), fabsf(t1)));   // (forcing v_and / v_max3 through inline asm saves 0.25 VALU per MFMA and costs
                                                 // 30 % of the kernel: asm statements are opaque to the MFMA / VALU interleave)
    if constexpr (HALF == 0) {
        pend[0] = t0, pend[1] = t1;
    } else {
        *reinterpret_cast<float4v*>(seg_row + 32 * ks + 8 * g + 4 * (RT & 1)) = float4v{pend[0], pend[1], t0, t1};
    }
    if constexpr (OPERAND) {
        unsigned hpk, lpk;
        split_pair_signed(v0 * s_op, v1 * s_op, hpk, lpk);
        set_pair(oh, e0, hpk);
        set_pair(ol, e0, lpk);
    }
}

__global__ void __launch_bounds__(CfgT::NWAVES * 64, 2) train_dgrad_kernel(BwdArgs a) {
    using C = CfgT;
    __shared__ __attribute__((aligned(16))) char smem[C::RING_BYTES + kNerfBiasBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, n = lane & 15;
    const long long tile = (long long)blockIdx.x * C::NWAVES + wave;
    const long long s = tile * 16 + n;
    const bool live = s < a.M;

    // ---- ordinary loads first: the ten gate words of this lane, the head gradients of its sample
    unsigned long long gw[kGateLayers];
#pragma unroll
    for (int l = 0; l < kGateLayers; ++l) gw[l] = a.gates[((size_t)l * a.n_tiles + tile) * 64 + lane];
    float hd[4] = {0.f, 0.f, 0.f, 0.f};   // [d sigma, dz_r, dz_g, dz_b] (rgb = sigmoid(z): dz = d rgb * rgb * (1 - rgb), models.py:111)
    if (live) {
        hd[0] = a.d_sigma[s];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float y = a.rgb[s * 3 + k];
            hd[1 + k] = a.d_rgb[s * 3 + k] * y * (1.0f - y);
        }
    }
#pragma unroll
    for (int l = 0; l < kGateLayers; ++l) asm volatile("" : "+v"(gw[l]));
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(hd[k]));

    WeightStream<C, SingleStreamMap<kDgradFrags>> ws;
    const char* const streams[1] = {a.stream};
    ws.init(streams, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < kNerfBiasBytes / (C::NWAVES * 1024); ++j)
        lds_dma16(a.zero_bias + (j * C::NWAVES + wave) * 1024 + lane * 16, smem + C::RING_BYTES + (j * C::NWAVES + wave) * 1024);
    ws.prologue();

    const long long m_pad = a.n_tiles * 16;               // rows up to M_pad exist; dead samples carry zeros
    auto seg_row = [&](int col0, int width) { return a.dz + seg_at(col0, width, m_pad, s); };
    // heads: true values to the stash (16 floats: [d sigma, dz rgb, 0 ...]), tile maximum -> first scale
    *reinterpret_cast<float4v*>(seg_row(Z_HEADS, 16) + 4 * g) = g == 0 ? float4v{hd[0], hd[1], hd[2], hd[3]} : float4v{0.f, 0.f, 0.f, 0.f};
    // segment maxima of this wave's tile: kept (wave-uniform) until the chain is through and published then -- an atomic is
    // a vector-memory operation, and one per layer in front of the ring's counted waits stalled every layer's first chunks
    unsigned seg_max[12];
    // The two heads carry their own scales (and their own segment maxima, 10: colour, 11: sigma): d sigma and dz_rgb may be
    // orders of magnitude apart (compositing gives |d sigma| ~ 1e-3 |d rgb| early in training and the reverse is as legal),
    // and the smaller one at the larger one's scale loses its low bits to fp16.  The colour head's values enter at D0, the
    // sigma row at D2 -- two layers after the colour branch has shrunk its values and the lagged scale has grown: no scale up to
    // and including D2's outputs may exceed the one d sigma fits in, or d sigma (and the products it feeds) leave the fp16
    // range: hi = inf, lo = x - inf, and the layer sums to NaN without any inf left for the maxima to show (round 4).
    const float m_sig = wave_max(fabsf(hd[0]));
    float m = wave_max(fmaxf(fabsf(hd[1]), fmaxf(fabsf(hd[2]), fabsf(hd[3]))));
    seg_max[10] = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, m));
    seg_max[11] = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, m_sig));
    const int e_sig = scale_exp(m_sig, 100);   // (no sigma gradient in this tile: no constraint)
    int e_in = scale_exp(m, e_sig);            // (no colour gradient: the branch carries zeros, at the scale sigma will need)
    // natural order k = 8g + j: the four values sit in lane group 0; SIGMA: the sigma entry alone (D2), else the colour entries (D0)
    auto heads_frag = [&](auto sigma_, int e, half8& h, half8& l) {
        constexpr bool SIGMA = decltype(sigma_)::value;
        const float sc = pow2f(e);
        unsigned h01 = 0, l01 = 0, h23 = 0, l23 = 0;
        if (g == 0) {
            split_pair_signed(SIGMA ? hd[0] * sc : 0.f, SIGMA ? 0.f : hd[1] * sc, h01, l01);
            split_pair_signed(SIGMA ? 0.f : hd[2] * sc, SIGMA ? 0.f : hd[3] * sc, h23, l23);
        }
        h = __builtin_bit_cast(half8, u4{h01, h23, 0u, 0u});
        l = __builtin_bit_cast(half8, u4{l01, l23, 0u, 0u});
    };

    const lds_cptr bias_lane = opaque((lds_cptr)smem + C::RING_BYTES + 16 * g);
    ws.start();

    half8 Xh[8][1], Xl[8][1], Yh[8][1], Yl[8][1];
    float pend[2];
    // after a layer: the tile maximum of its gated outputs fixes the scale of the layer AFTER the next (the next layer's
    // operands were already produced at the scale derived one layer earlier); segment maxima feed the weight-gradient kernel
    auto close = [&](auto seg_, float& mm, int cur, int keep) {   // cur: the scale in force, keep: the answer for an all-zero tile
        const float t = wave_max(mm);
        seg_max[decltype(seg_)::value] = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, t));
#ifdef TGTC_DGRAD_ATOMIC_PER_LAYER   // A/B build: the round's first form, one atomic per layer in front of the ring waits
        if (lane == 0) atomicMax(a.maxima + decltype(seg_)::value, __builtin_bit_cast(unsigned, t));
#endif
        mm = 0.f;
        return max(cur - 120, min(cur + 120, scale_exp(t, keep)));   // (s_op = 2^(difference of consecutive scales) stays a normal float)
    };

    // D0: rgb_layers.1^T.  inputs heads (scale e_in), outputs dz_f at the same scale (|W| < 1: the values shrink)
    int e_out = e_in, e_next;
    {
        half8 Bh[1][1], Bl[1][1];
        heads_frag(std::false_type{}, e_in, Bh[0][0], Bl[0][0]);
        half8 Zh[4][1], Zl[4][1];
        m = 0.f;
        {
            const float s_true = pow2f(-e_in), s_op = pow2f(e_out - e_in);
            float* const zrow = seg_row(Z_F, 128);
            dense_layer<C, dgrad_frag0(0), 1, 8, 0>(ws, bias_lane, Bh, Bl, [&](auto rt_, auto, auto h_, const float4v& acc) {
                constexpr int rt = decltype(rt_)::value, hf = decltype(h_)::value;
                dgrad_epi<rt, hf, true>(acc, gw[9], s_true, s_op, m, pend, zrow, g, Zh[rt / 2][0], Zl[rt / 2][0]);
            });
        }
        e_next = min(close(ic<9>{}, m, e_out, e_sig), e_sig);
        // D1: rgb_layers.0^T (activation columns): dz_f -> dz_remap
        e_in = e_out, e_out = e_next;
        {
            const float s_true = pow2f(-e_in), s_op = pow2f(e_out - e_in);
            float* const zrow = seg_row(Z_REMAP, 256);
            dense_layer<C, dgrad_frag0(1), 4, 16, 0>(ws, bias_lane, Zh, Zl, [&](auto rt_, auto, auto h_, const float4v& acc) {
                constexpr int rt = decltype(rt_)::value, hf = decltype(h_)::value;
                dgrad_epi<rt, hf, true>(acc, gw[8], s_true, s_op, m, pend, zrow, g, Yh[rt / 2][0], Yl[rt / 2][0]);
            });
        }
        e_next = min(close(ic<8>{}, m, e_out, e_sig), e_sig);
    }
    // D2: [base_remap^T | sigma^T]: [dz_remap | heads] -> dz_7
    e_in = e_out, e_out = e_next;
    {
        half8 Bh[9][1], Bl[9][1];
#pragma unroll
        for (int k = 0; k < 8; ++k) Bh[k][0] = Yh[k][0], Bl[k][0] = Yl[k][0];
        heads_frag(std::true_type{}, e_in, Bh[8][0], Bl[8][0]);
        const float s_true = pow2f(-e_in), s_op = pow2f(e_out - e_in);
        float* const zrow = seg_row(z_layer(7), 256);
        dense_layer<C, dgrad_frag0(2), 9, 16, 0>(ws, bias_lane, Bh, Bl, [&](auto rt_, auto, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, hf = decltype(h_)::value;
            dgrad_epi<rt, hf, true>(acc, gw[7], s_true, s_op, m, pend, zrow, g, Xh[rt / 2][0], Xl[rt / 2][0]);
        });
    }
    e_next = close(ic<7>{}, m, e_out, e_out);
    // D3..D9: base_layers[7..1]^T: dz_l -> dz_{l-1}
    auto hidden = [&](auto d_, auto last_, const half8 (&Ih)[8][1], const half8 (&Il)[8][1], half8 (&Oh)[8][1], half8 (&Ol)[8][1]) {
        constexpr int d = decltype(d_)::value, l_out = 9 - d;          // D3 -> dz_6 ... D9 -> dz_0
        constexpr bool last = decltype(last_)::value;
        e_in = e_out, e_out = e_next;
        const float s_true = pow2f(-e_in), s_op = pow2f(e_out - e_in);
        float* const zrow = seg_row(z_layer(l_out), 256);
        dense_layer<C, dgrad_frag0(d), 8, 16, 0>(ws, bias_lane, Ih, Il, [&](auto rt_, auto, auto h_, const float4v& acc) {
            constexpr int rt = decltype(rt_)::value, hf = decltype(h_)::value;
            dgrad_epi<rt, hf, !last>(acc, gw[l_out], s_true, s_op, m, pend, zrow, g, Oh[rt / 2][0], Ol[rt / 2][0]);
        });
        e_next = close(ic<l_out>{}, m, e_out, e_out);
    };
    hidden(ic<3>{}, std::false_type{}, Xh, Xl, Yh, Yl);
    hidden(ic<4>{}, std::false_type{}, Yh, Yl, Xh, Xl);
    hidden(ic<5>{}, std::false_type{}, Xh, Xl, Yh, Yl);
    hidden(ic<6>{}, std::false_type{}, Yh, Yl, Xh, Xl);
    hidden(ic<7>{}, std::false_type{}, Xh, Xl, Yh, Yl);
    hidden(ic<8>{}, std::false_type{}, Yh, Yl, Xh, Xl);
    hidden(ic<9>{}, std::true_type{}, Xh, Xl, Yh, Yl);
    // a scaled operand beyond the fp16 range produces infinities, which reach the maxima (fmaxf drops a NaN, not an inf):
    // report instead of returning garbage
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 12; ++i) bad |= (seg_max[i] & 0x7f800000u) == 0x7f800000u;
    // lane i publishes segment i, and only if it raises the table (one coherent load + one masked atomic per wave; after the
    // first few tiles almost no lane has anything to add.  One unconditional atomic per wave and segment -- 90 000 on eleven
    // addresses -- cost 0.2 ms of the kernel wherever they were issued)
    unsigned mine = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) mine = lane == i ? seg_max[i] : mine;
    if (lane < 12 && mine > __hip_atomic_load(a.maxima + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.maxima + lane, mine);
    if (bad && lane == 0) atomicMax(a.status, 1u);
}

}  // namespace train
}  // namespace tgtc

// ------------------------------------------------------------------------------------------------ weight gradients
// dW[n, k] = sum over samples m of dZ[m, n] * Hin[m, k] for the twelve linears in ONE launch: a workgroup owns one job
// (train_layouts.h kWgradJob) and a contiguous range of 32-sample steps, keeps the WHOLE n x k tile set of the job in
// accumulators (8 waves x 1 row tile x up to 20 column tiles), stages the step's rows through LDS (dZ: fp32 ->
// scaled fp16 hi/lo; H: the stashed planes) and reads both MFMA operands TRANSPOSED out of the row-major images with
// ds_read_b64_tr_b16 (the reduction index -- the sample -- must run along the k dimension of both operands).  Partial tiles are
// added into the fp32 gradients with atomics (un-permuting both indices on the way); db = column sums of dZ ride along.
namespace tgtc {
namespace train {

struct WgradArgs {
    const half_t* h_hi;
    const half_t* h_lo;
    const float* dz;
    const unsigned* maxima;          // per dZ segment (see BwdArgs)
    long long steps;                 // M_pad / 32
    long long m_pad;
    const short* unperm;             // per job: [128] row map then [320] column map (logical index or -1)
    float* grads[24];                // dW (2 * layer) and db (2 * layer + 1) of the twelve linears, zero-filled by the caller
    int chunk0[kWgradJobs + 1];      // workgroup ranges of the jobs
};

constexpr int kWgRowPad = 8;         // halves of padding per LDS image row (16 bytes)
constexpr int kWgMaxN = 128, kWgMaxK = 320;
constexpr int wg_stride(int cols) { return cols + kWgRowPad; }          // halves
constexpr int kWgLdsBytes = 32 * (wg_stride(kWgMaxN) + wg_stride(kWgMaxK)) * 2 * 2;   // two planes each
static_assert(kWgLdsBytes <= 160 * 1024, "LDS");

__device__ __forceinline__ int seg_of_zcol(int z_col) {
    return z_col == Z_HEADS ? 10 : z_col == Z_F ? 9 : z_col == Z_REMAP ? 8 : z_col / 256;
}

// 4 samples x 16 features block of a row-major [sample][feature] fp16 image, transposed: lane i of each 16-lane group gets
// feature i of the 4 samples (cdna_hip_programming.md T10).  `p` = this lane's address: row q = (lane >> 2) & 3 of the block,
// features 4 * (lane & 3) .. + 3.
typedef unsigned u2w __attribute__((ext_vector_type(2)));
// The reads are inline asm (no builtin): hipcc neither counts them nor knows when their data lands (cdna_hip_programming.md
// section 5.7).  tr_issue starts the four reads of one operand pair (hi and lo plane, two 4-sample blocks each); tr_wait is the
// ONLY consumer-side statement that names their destinations before the MFMAs do, and carries the s_waitcnt.
struct TrQuad {
    u2w h0, h1, l0, l1;
};
__device__ __forceinline__ void tr_issue(TrQuad& r, const half_t* hi, const half_t* lo, int stride4) {
    const unsigned ah = (unsigned)(size_t)TGTC_LPTR(hi), al = (unsigned)(size_t)TGTC_LPTR(lo);
    const unsigned bh = ah + 2 * stride4, bl = al + 2 * stride4;     // the block of the next four samples (byte offset)
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7"
                 : "=&v"(r.h0), "=&v"(r.h1), "=&v"(r.l0), "=&v"(r.l1)
                 : "v"(ah), "v"(bh), "v"(al), "v"(bl)
                 : "memory");
}
__device__ __forceinline__ void tr_wait(TrQuad& r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.h0), "+v"(r.h1), "+v"(r.l0), "+v"(r.l1) : : "memory");
}
__device__ __forceinline__ half8 tr_hi(const TrQuad& r) { return __builtin_bit_cast(half8, u4{r.h0[0], r.h0[1], r.h1[0], r.h1[1]}); }
__device__ __forceinline__ half8 tr_lo(const TrQuad& r) { return __builtin_bit_cast(half8, u4{r.l0[0], r.l0[1], r.l1[0], r.l1[1]}); }

// N rows of dW (a dZ slice N wide), K0 + K1 input columns (two H segments), all compile-time: every staging load is
// unconditional and issued back to back (a load behind a lane-dependent branch made hipcc wait for the previous one first:
// three memory round trips per step), out-of-range threads re-read a valid unit and skip the LDS store.
template <int N, int K0, int K1>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const WgradJob& J, int job, long long step0, long long step1, char* smem) {
    constexpr int KT = (K0 + K1) / 16, NT = N / 16;
    constexpr int SA = wg_stride(N), SB = wg_stride(K0 + K1);   // halves per image row
    constexpr int A_UPR = N / 4, A_UNITS = 32 * A_UPR, A_PER = (A_UNITS + 511) / 512;      // 16-byte units: per row, per step, per thread
    constexpr int B0_UPR = K0 / 8, B0_UNITS = 32 * B0_UPR, B0_PER = (B0_UNITS + 511) / 512;
    constexpr int B1_UPR = K1 ? K1 / 8 : 1, B1_UNITS = K1 ? 32 * B1_UPR : 0;
    static_assert(B1_UNITS <= 512 && A_PER <= 2 && B0_PER <= 2, "staging plan");
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    half_t* const Ahi = reinterpret_cast<half_t*>(smem);
    half_t* const Alo = Ahi + 32 * SA;
    half_t* const Bhi = Alo + 32 * SA;
    half_t* const Blo = Bhi + 32 * SB;

    // scale of the dZ operand: a power of two that brings the segment maximum to ~2^10
    const unsigned mb = a.maxima[J.z_col == Z_HEADS && J.layer == 8 ? 11 : seg_of_zcol(J.z_col)];   // (the sigma row has its own maximum)
    const int e_sc = mb == 0 ? 0 : max(-100, min(100, 10 - ((int)(mb >> 23) - 127)));   // (bounded like scale_exp)
    const float sc = pow2f(e_sc);

    float4v acc[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[j] = float4v{0.f, 0.f, 0.f, 0.f};
    float4v bsum = float4v{0.f, 0.f, 0.f, 0.f};

    // per-thread source pointers of step `step0`, advanced by 32 rows per step
    const float* pa[A_PER];
    const half_t *pb0h[B0_PER], *pb0l[B0_PER], *pb1h = nullptr, *pb1l = nullptr;
    int la[A_PER], lb0[B0_PER], lb1 = 0;       // LDS offsets (halves); -1: this thread has no such unit
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int u = t + 512 * i, ue = u < A_UNITS ? u : u % A_UNITS;
        pa[i] = a.dz + seg_at(J.z_col, J.z_n, a.m_pad, step0 * 32 + ue / A_UPR) + J.n0 + 4 * (ue % A_UPR);
        la[i] = u < A_UNITS ? (u / A_UPR) * SA + 4 * (u % A_UPR) : -1;
    }
#pragma unroll
    for (int i = 0; i < B0_PER; ++i) {
        const int u = t + 512 * i, ue = u < B0_UNITS ? u : u % B0_UNITS;
        const size_t off = seg_at(J.k_col[0], K0, a.m_pad, step0 * 32 + ue / B0_UPR) + 8 * (ue % B0_UPR);
        pb0h[i] = a.h_hi + off, pb0l[i] = a.h_lo + off;
        lb0[i] = u < B0_UNITS ? (u / B0_UPR) * SB + 8 * (u % B0_UPR) : -1;
    }
    if constexpr (K1 > 0) {
        const int ue = t % B1_UNITS;
        const size_t off = seg_at(J.k_col[1], K1, a.m_pad, step0 * 32 + ue / B1_UPR) + 8 * (ue % B1_UPR);
        pb1h = a.h_hi + off, pb1l = a.h_lo + off;
        lb1 = t < B1_UNITS ? (t / B1_UPR) * SB + K0 + 8 * (t % B1_UPR) : -1;
    }
    float4v ra[A_PER];
    u4 rb0h[B0_PER], rb0l[B0_PER], rb1h = u4{0, 0, 0, 0}, rb1l = u4{0, 0, 0, 0};
    auto load_step = [&]() {    // the step the pointers stand on; then move them on
#pragma unroll
        for (int i = 0; i < A_PER; ++i) ra[i] = *reinterpret_cast<const float4v*>(pa[i]), pa[i] += 32 * (size_t)J.z_n;
#pragma unroll
        for (int i = 0; i < B0_PER; ++i) {
            rb0h[i] = *reinterpret_cast<const u4*>(pb0h[i]), rb0l[i] = *reinterpret_cast<const u4*>(pb0l[i]);
            pb0h[i] += 32 * K0, pb0l[i] += 32 * K0;
        }
        if constexpr (K1 > 0) {
            rb1h = *reinterpret_cast<const u4*>(pb1h), rb1l = *reinterpret_cast<const u4*>(pb1l);
            pb1h += 32 * K1, pb1l += 32 * K1;
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            if (la[i] >= 0) {
                const float4v v = ra[i];
                bsum += v;
                unsigned h01, l01, h23, l23;
                split_pair_signed(v[0] * sc, v[1] * sc, h01, l01);
                split_pair_signed(v[2] * sc, v[3] * sc, h23, l23);
                *reinterpret_cast<u2w*>(Ahi + la[i]) = u2w{h01, h23};
                *reinterpret_cast<u2w*>(Alo + la[i]) = u2w{l01, l23};
            }
        }
#pragma unroll
        for (int i = 0; i < B0_PER; ++i)
            if (lb0[i] >= 0) *reinterpret_cast<u4*>(Bhi + lb0[i]) = rb0h[i], *reinterpret_cast<u4*>(Blo + lb0[i]) = rb0l[i];
        if constexpr (K1 > 0)
            if (lb1 >= 0) *reinterpret_cast<u4*>(Bhi + lb1) = rb1h, *reinterpret_cast<u4*>(Blo + lb1) = rb1l;
    };

    // transposed-read addresses: lane group G = lane >> 4 takes samples 8G .. 8G+7 (two blocks of 4), lane 4q + p of the
    // group points at row q of the block, features 4p .. 4p+3 of the 16-feature tile
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int nt = wave < NT ? wave : 0;        // idle waves read tile 0 and discard (EXEC stays full for the reads)
    const half_t* const a_hi = Ahi + (8 * G + q) * SA + 4 * p + 16 * nt;
    const half_t* const a_lo = Alo + (8 * G + q) * SA + 4 * p + 16 * nt;
    const int b_row = (8 * G + q) * SB + 4 * p;

    load_step();
    for (long long step = step0; step < step1; ++step) {
        __syncthreads();          // everyone is done reading the previous step's images
        store_step();
        __syncthreads();
        if (step + 1 < step1) load_step();      // in flight while this step multiplies
        TrQuad qa, qb[2];
        tr_issue(qa, a_hi, a_lo, 4 * SA);
        tr_issue(qb[0], Bhi + b_row, Blo + b_row, 4 * SB);
        tr_wait(qa);
        const half8 ah = tr_hi(qa), al = tr_lo(qa);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            tr_wait(qb[kt & 1]);
            const half8 bh = tr_hi(qb[kt & 1]), bl = tr_lo(qb[kt & 1]);
            if (kt + 1 < KT) tr_issue(qb[(kt + 1) & 1], Bhi + b_row + 16 * (kt + 1), Blo + b_row + 16 * (kt + 1), 4 * SB);
            __builtin_amdgcn_sched_barrier(0);
            acc[kt] = mfma16(ah, bh, acc[kt]);
            acc[kt] = mfma16(al, bh, acc[kt]);
            acc[kt] = mfma16(ah, bl, acc[kt]);
        }
    }

    // ---- add the partial tiles into the gradients: row 4 * (lane >> 4) + r of the tile, column lane & 15
    const short* const nmap = a.unperm + job * (kWgMaxN + kWgMaxK);
    const short* const kmap = nmap + kWgMaxN;
    float* const dW = a.grads[2 * J.layer];
    float* const db = a.grads[2 * J.layer + 1];
    const int in_features = J.layer == 0 ? 63 : J.layer == 5 ? 319 : J.layer == 10 ? 283 : J.layer == 11 ? 128 : 256;
    const float inv = pow2f(-e_sc);
    if (wave < NT) {
        int nr[4], kc[KT];          // all map entries first (independent loads), then the atomics
#pragma unroll
        for (int r = 0; r < 4; ++r) nr[r] = nmap[16 * wave + 4 * (lane >> 4) + r];   // (the job's map already starts at its n0)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) kc[kt] = kmap[16 * kt + (lane & 15)];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (nr[r] >= 0 && kc[kt] >= 0) atomicAdd(dW + (size_t)nr[r] * in_features + kc[kt], acc[kt][r] * inv);
    }
    // db: this thread's four dZ columns (the same in every step)
    if (la[0] >= 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nr = nmap[4 * (t % A_UPR) + i];
            if (nr >= 0) atomicAdd(db + nr, bsum[i]);
        }
    }
}

__global__ void __launch_bounds__(512, 2) train_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int job = 0;
#pragma unroll
    for (int j = 1; j < kWgradJobs; ++j)
        if ((int)blockIdx.x >= a.chunk0[j]) job = j;
    const int nchunk = a.chunk0[job + 1] - a.chunk0[job], chunk = (int)blockIdx.x - a.chunk0[job];
    const long long step0 = a.steps * chunk / nchunk, step1 = a.steps * (chunk + 1) / nchunk;
    const WgradJob J = kWgradJob[job];
    switch (J.layer) {
        case 0: wgrad_body<128, 64, 0>(a, J, job, step0, step1, smem); break;
        case 5: wgrad_body<128, 256, 64>(a, J, job, step0, step1, smem); break;
        case 8: wgrad_body<16, 256, 0>(a, J, job, step0, step1, smem); break;
        case 10: wgrad_body<128, 256, 32>(a, J, job, step0, step1, smem); break;
        case 11: wgrad_body<16, 128, 0>(a, J, job, step0, step1, smem); break;
        default: wgrad_body<128, 256, 0>(a, J, job, step0, step1, smem); break;
    }
}

}  // namespace train
}  // namespace tgtc

// ------------------------------------------------------------------------------------------------ C ABI
using namespace tgtc;
using namespace tgtc::train;

namespace {

int frag_order_feature(int kind, int c) {   // logical index of fragment-order column c of a segment (-1: padding)
    const int ks = c / 32, g = (c % 32) / 8, j = c % 8;
    switch (kind) {
        case 0: return act_col(ks, g, j);
        case 1: return pe63_col(ks, g, j);
        default: return pe27_col(g, j);
    }
}

void build_unperm(std::vector<short>& out) {
    out.assign((size_t)kWgradJobs * (kWgMaxN + kWgMaxK), (short)-1);
    for (int job = 0; job < kWgradJobs; ++job) {
        const WgradJob& J = kWgradJob[job];
        short* nmap = out.data() + (size_t)job * (kWgMaxN + kWgMaxK);
        short* kmap = nmap + kWgMaxN;
        for (int c = 0; c < J.n; ++c) {
            if (J.z_col == Z_HEADS) nmap[c] = (short)(J.layer == 8 ? (c == 0 ? 0 : -1) : (c >= 1 && c <= 3 ? c - 1 : -1));
            else nmap[c] = (short)frag_order_feature(0, J.n0 + c);
        }
        int at = 0;
        for (int s = 0; s < 2; ++s)
            for (int c = 0; c < J.k_n[s]; ++c, ++at) {
                int f;
                if (J.k_col[s] == H_PE) f = frag_order_feature(1, c);                       // the 63 encoding columns come first
                else if (J.k_col[s] == H_DIR) f = frag_order_feature(2, c) < 0 ? -1 : 256 + frag_order_feature(2, c);
                else f = frag_order_feature(0, c) + (J.layer == 5 ? 63 : 0);               // cat(pe, h): h behind the 63 pe columns
                kmap[at] = (short)f;
            }
    }
}

}  // namespace

extern "C" int tgtc_trainer_create(tgtc_trainer** out) {
    TGTC_REQUIRE(out, "trainer_create: null argument");
    std::vector<PackSrc> fwd, bwd, bias;
    build_stream_map(forward_layers(), fwd);
    build_stream_map(dgrad_layers(), bwd);
    if ((int)fwd.size() < NerfLayout::kFragsFull * 512 || (int)bwd.size() < kDgradFrags * 512)
        return fail(TGTC_ERR_UNSUPPORTED, "trainer_create: internal stream map mismatch (%zu, %zu slots)", fwd.size(), bwd.size());
    static const int outs[12] = {256, 256, 256, 256, 256, 256, 256, 256, 1, 256, 128, 3};
    bias.assign(kNerfBiasBytes / 4, PackSrc{-1, 0});
    for (int l = 0, b0 = 0; l < 12; ++l) {
        for (int r = 0; r < outs[l]; ++r) bias[b0 + r] = PackSrc{2 * l + 1, r};
        b0 += 16 * ((outs[l] + 15) / 16);
    }
    std::vector<short> unperm;
    build_unperm(unperm);
    tgtc_trainer* tr = new tgtc_trainer();
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    };
    tr->fwd_slots = (int)fwd.size(), tr->bwd_slots = (int)bwd.size();
    tr->fwd_stream_bytes = fwd.size() * 4, tr->bwd_stream_bytes = bwd.size() * 4;      // hi + lo = 4 bytes per slot
    tr->fwd_map_off = take(fwd.size() * sizeof(PackSrc));
    tr->bwd_map_off = take(bwd.size() * sizeof(PackSrc));
    tr->fwd_bias_map_off = take(bias.size() * sizeof(PackSrc));
    tr->unperm_off = take(unperm.size() * sizeof(short));
    tr->maxima_off = take(64);                                                         // 12 segment maxima + status word
    tr->fwd_stream_off = take(kNerfBiasBytes + tr->fwd_stream_bytes + kRingBytes);     // [bias table][stream][slack for the look-ahead]
    tr->bwd_stream_off = take(kNerfBiasBytes + tr->bwd_stream_bytes + kRingBytes);     // [zeros][stream][slack]
    hipError_t e = hipMalloc((void**)&tr->dev, off);
    if (e != hipSuccess) {
        delete tr;
        return fail(TGTC_ERR_HIP, "trainer_create: hipMalloc(%zu): %s", off, hipGetErrorString(e));
    }
    std::vector<char> host(off, 0);
    std::memcpy(host.data() + tr->fwd_map_off, fwd.data(), fwd.size() * sizeof(PackSrc));
    std::memcpy(host.data() + tr->bwd_map_off, bwd.data(), bwd.size() * sizeof(PackSrc));
    std::memcpy(host.data() + tr->fwd_bias_map_off, bias.data(), bias.size() * sizeof(PackSrc));
    std::memcpy(host.data() + tr->unperm_off, unperm.data(), unperm.size() * sizeof(short));
    e = hipMemcpy(tr->dev, host.data(), off, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(tr->dev);
        delete tr;
        return fail(TGTC_ERR_HIP, "trainer_create: hipMemcpy: %s", hipGetErrorString(e));
    }
    *out = tr;
    return TGTC_OK;
}

// Overflow guard: the input-gradient chain raises `status` when a scaled operand left the fp16 range -- its
// gradients are then inf / NaN, and an optimiser step would destroy the weights for good.  Runs last in every backward:
// if the flag is set it zero-fills all 24 gradient tensors (the step becomes a no-op apart from the optimiser's momentum)
// and counts the event in a word that survives the per-call reset (tgtc_trainer_overflows).
struct GuardArgs {
    float* g[24];
    unsigned n[24];
};
// (first pass of the guard) any non-finite gradient raises the flag too, whatever produced it: a forward that overflowed fp16,
// non-finite dL/d outputs -- fmaxf drops NaNs, so the chain's own maxima cannot see those.  2.4 MB of reads.
__global__ void __launch_bounds__(256) nonfinite_scan_kernel(unsigned* status, GuardArgs a) {
    bool bad = false;
    for (int i = 0; i < 24; ++i)
        for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < a.n[i]; k += gridDim.x * blockDim.x)
            bad |= (__builtin_bit_cast(unsigned, a.g[i][k]) & 0x7f800000u) == 0x7f800000u;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(status, 1u);
}
__global__ void __launch_bounds__(256) overflow_guard_kernel(const unsigned* status, unsigned* count, GuardArgs a) {
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    for (int i = 0; i < 24; ++i)
        for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < a.n[i]; k += gridDim.x * blockDim.x) a.g[i][k] = 0.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(count, 1u);
}

extern "C" int tgtc_trainer_destroy(tgtc_trainer* tr) {
    if (!tr) return TGTC_OK;
    const hipError_t e = tr->dev ? hipFree(tr->dev) : hipSuccess;
    delete tr;
    if (e != hipSuccess) return fail(TGTC_ERR_HIP, "trainer_destroy: hipFree: %s", hipGetErrorString(e));
    return TGTC_OK;
}

extern "C" size_t tgtc_trainer_workspace_bytes(int64_t M) { return M > 0 ? ws_layout(M).total : 0; }

static int check_params(const float* const* params, const char* who) {
    TGTC_REQUIRE(params, "%s: null parameter table", who);
    for (int i = 0; i < 24; ++i) TGTC_REQUIRE(params[i], "%s: parameter %d is null", who, i);
    return TGTC_OK;
}

extern "C" int tgtc_trainer_forward(tgtc_trainer* tr, const float* const* params, const double* pts, const double* dirs, int64_t M,
                                    void* workspace, size_t workspace_bytes, float* rgb, float* sigma, void* stream) {
    TGTC_REQUIRE(tr && pts && dirs && workspace && rgb && sigma, "trainer_forward: null argument");
    TGTC_REQUIRE(M > 0 && M < 0x7fffffffLL / 32, "trainer_forward: M = %lld out of range", (long long)M);
    if (int rc = check_params(params, "trainer_forward")) return rc;
    const WsLayout w = ws_layout(M);
    TGTC_REQUIRE(workspace_bytes >= w.total, "trainer_forward: workspace of %zu bytes, need %zu", workspace_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    ParamPtrs pp;
    for (int i = 0; i < 24; ++i) pp.p[i] = params[i];
    char* fs = tr->dev + tr->fwd_stream_off;
    gather_pack_kernel<<<(tr->fwd_slots + 255) / 256, 256, 0, st>>>(pp, reinterpret_cast<const PackSrc*>(tr->dev + tr->fwd_map_off), tr->fwd_slots,
                                                                    reinterpret_cast<half_t*>(fs + kNerfBiasBytes),
                                                                    reinterpret_cast<const PackSrc*>(tr->dev + tr->fwd_bias_map_off),
                                                                    kNerfBiasBytes / 4, reinterpret_cast<float*>(fs));
    TGTC_LAUNCH_CHECK();
    char* ws = static_cast<char*>(workspace);
    FwdArgs a{};
    a.bias = fs, a.stream = fs + kNerfBiasBytes, a.M = M, a.pts = pts, a.dirs = dirs;
    a.h_hi = reinterpret_cast<half_t*>(ws + w.h_hi), a.h_lo = reinterpret_cast<half_t*>(ws + w.h_lo);
    a.gates = reinterpret_cast<unsigned long long*>(ws + w.gates), a.n_tiles = w.n_tiles, a.rgb = rgb, a.sigma = sigma;
    train_forward_kernel<<<(unsigned)(w.m_pad / kTileSamples), CfgT::NWAVES * 64, 0, st>>>(a);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_trainer_backward(tgtc_trainer* tr, const float* const* params, const float* rgb, const float* d_rgb,
                                     const float* d_sigma, int64_t M, void* workspace, size_t workspace_bytes, float* const* grads,
                                     void* stream) {
    TGTC_REQUIRE(tr && rgb && d_rgb && d_sigma && workspace && grads, "trainer_backward: null argument");
    TGTC_REQUIRE(M > 0 && M < 0x7fffffffLL / 32, "trainer_backward: M = %lld out of range", (long long)M);
    if (int rc = check_params(params, "trainer_backward")) return rc;
    for (int i = 0; i < 24; ++i) TGTC_REQUIRE(grads[i], "trainer_backward: gradient %d is null", i);
    const WsLayout w = ws_layout(M);
    TGTC_REQUIRE(workspace_bytes >= w.total, "trainer_backward: workspace of %zu bytes, need %zu", workspace_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    ParamPtrs pp;
    for (int i = 0; i < 24; ++i) pp.p[i] = params[i];
    char* bs = tr->dev + tr->bwd_stream_off;    // its first kNerfBiasBytes stay zero: the transposed products carry no bias
    gather_pack_kernel<<<(tr->bwd_slots + 255) / 256, 256, 0, st>>>(pp, reinterpret_cast<const PackSrc*>(tr->dev + tr->bwd_map_off), tr->bwd_slots,
                                                                    reinterpret_cast<half_t*>(bs + kNerfBiasBytes), nullptr, 0, nullptr);
    TGTC_LAUNCH_CHECK();
    unsigned* maxima = reinterpret_cast<unsigned*>(tr->dev + tr->maxima_off);
    TGTC_HIP_CHECK(hipMemsetAsync(maxima, 0, 52, st));    // 12 segment maxima, status word at +48; the overflow counter at +52 survives
    static const int shape[12][2] = {{256, 63}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 319}, {256, 256}, {256, 256},
                                     {1, 256},  {256, 256}, {128, 283}, {3, 128}};
    {   // the weight-gradient kernel accumulates: zero-fill first (ONE fill when the caller laid the 24 tensors out back to back)
        size_t bytes[24], total = 0;
        bool packed = true;
        for (int l = 0; l < 12; ++l) bytes[2 * l] = (size_t)shape[l][0] * shape[l][1] * sizeof(float), bytes[2 * l + 1] = (size_t)shape[l][0] * sizeof(float);
        for (int i = 0; i < 24; ++i) {
            if (i + 1 < 24 && reinterpret_cast<char*>(grads[i]) + bytes[i] != reinterpret_cast<char*>(grads[i + 1])) packed = false;
            total += bytes[i];
        }
        if (packed) {
            TGTC_HIP_CHECK(hipMemsetAsync(grads[0], 0, total, st));
        } else {
            for (int i = 0; i < 24; ++i) TGTC_HIP_CHECK(hipMemsetAsync(grads[i], 0, bytes[i], st));
        }
    }
    char* ws = static_cast<char*>(workspace);
    BwdArgs b{};
    b.zero_bias = bs, b.stream = bs + kNerfBiasBytes, b.M = M, b.rgb = rgb, b.d_rgb = d_rgb, b.d_sigma = d_sigma;
    b.gates = reinterpret_cast<const unsigned long long*>(ws + w.gates), b.n_tiles = w.n_tiles;
    b.dz = reinterpret_cast<float*>(ws + w.dz), b.maxima = maxima, b.status = maxima + 12;
    train_dgrad_kernel<<<(unsigned)(w.m_pad / kTileSamples), CfgT::NWAVES * 64, 0, st>>>(b);
    TGTC_LAUNCH_CHECK();

    WgradArgs g{};
    g.h_hi = reinterpret_cast<const half_t*>(ws + w.h_hi), g.h_lo = reinterpret_cast<const half_t*>(ws + w.h_lo);
    g.dz = reinterpret_cast<const float*>(ws + w.dz), g.maxima = maxima, g.steps = w.m_pad / 32, g.m_pad = w.m_pad;
    g.unperm = reinterpret_cast<const short*>(tr->dev + tr->unperm_off);
    for (int i = 0; i < 24; ++i) g.grads[i] = grads[i];
    // A step costs every job about the same (it is bound by the round trip of its staging loads, not by its 12 .. 60 MFMAs
    // per wave), so every job gets the same number of sample chunks: 21 jobs x 12 = 252 workgroups, one round on 256 CUs.
    // (In proportion to the multiply count the two one-tile jobs ran all 4 096 steps in ONE workgroup: 4 ms.)
    g.chunk0[0] = 0;
    for (int j = 0; j < kWgradJobs; ++j) g.chunk0[j + 1] = g.chunk0[j] + (int)std::min<long long>(12, g.steps);
    TGTC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(train_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kWgLdsBytes));
    train_wgrad_kernel<<<(unsigned)g.chunk0[kWgradJobs], 512, kWgLdsBytes, st>>>(g);
    TGTC_LAUNCH_CHECK();
    GuardArgs ga{};
    for (int l = 0; l < 12; ++l) {
        ga.g[2 * l] = grads[2 * l], ga.n[2 * l] = (unsigned)(shape[l][0] * shape[l][1]);
        ga.g[2 * l + 1] = grads[2 * l + 1], ga.n[2 * l + 1] = (unsigned)shape[l][0];
    }
#ifndef TGTC_ABL_NOSCAN   // (development builds can leave the scan out to see which of the two raised the flag)
    nonfinite_scan_kernel<<<64, 256, 0, st>>>(maxima + 12, ga);
    TGTC_LAUNCH_CHECK();
#endif
    overflow_guard_kernel<<<64, 256, 0, st>>>(maxima + 12, maxima + 13, ga);
    TGTC_LAUNCH_CHECK();
    return TGTC_OK;
}

extern "C" int tgtc_trainer_overflows(tgtc_trainer* tr, void* stream, unsigned* count) {
    TGTC_REQUIRE(tr && count, "trainer_overflows: null argument");
    TGTC_HIP_CHECK(hipMemcpyAsync(count, tr->dev + tr->maxima_off + 52, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    TGTC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return TGTC_OK;
}

extern "C" int tgtc_trainer_status(tgtc_trainer* tr, void* stream) {
    TGTC_REQUIRE(tr, "trainer_status: null argument");
    unsigned s = 0;
    TGTC_HIP_CHECK(hipMemcpyAsync(&s, tr->dev + tr->maxima_off + 48, sizeof(s), hipMemcpyDeviceToHost, (hipStream_t)stream));
    TGTC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (s) return fail(TGTC_ERR_UNSUPPORTED, "trainer: a scaled gradient left the fp16 range in the last backward, or a gradient came out non-finite (the gradients were zero-filled)");
    return TGTC_OK;
}
