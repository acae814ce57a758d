// Layouts of the fused TRAINING path of the NeRF MLP (mlp_train.hip; reference train_tgtcs.py:218-309 Origin_train
// backpropagates through MLP_style, models.py:95-117): what the forward kernel leaves in HBM for the backward kernels.
//
// Every feature set (segment) is its own [M_pad][width] array, one behind the other (segment-major: a producer writes, and a
// weight-gradient job reads, contiguous memory; with whole-network rows each 512-byte piece sat in its own 10 KB row and
// the gradient kernel ran at a fifth of the memory rate).  The "column" constants below are the prefix sums of the segment
// widths: segment s of a plane starts at element  col(s) * M_pad,  sample m of it at  + m * width(s).
// Within a segment the columns are in the "fragment order" of a 256-wide (or 128 / 64 / 32-wide) feature set: column
// c' = 32*ks + 8*g + j holds the feature that sits in element j of lane group g of k-step ks of a B fragment
// (mlp_core.h act_col / pe63_col / pe27_col).  In that order a lane's eight values of a k-step are 16 contiguous bytes of
// its sample's row, four lane groups make 64 contiguous bytes, and the weight-gradient kernel un-permutes both indices of
// dW once at the end (65 536 values per layer) instead of every producer permuting millions of activations.
//
//   H stash   fp16, two planes (hi, lo = the operands of the fp16x3 products), segments:
//             [pe(64) | h0 .. h7 (8 x 256) | base_remap (256) | f = relu(rgb_layers.0) (128) | dirs(32)]
//   dZ stash  fp32, segments: gradients w.r.t. the pre-activations, ReLU-gated, TRUE scale:
//             [dz0 .. dz7 (8 x 256) | dz_remap (256) | dz_f (128) | heads (16: d sigma, d rgb pre-sigmoid x3, zeros)]
//   gates     64 bits per (gated layer, sample tile, lane): bit 8*ks + (j >> 1) + 4*(j & 1) set iff element j of k-step ks of
//             the lane's activation fragment is positive.  Gated layers: h0..h7 (0..7), base_remap (8), f (9).
#pragma once
#include "mlp_core.h"
#include "mlp_layouts.h"

namespace tgtc {
namespace train {

constexpr int H_PE = 0;
constexpr int h_layer(int l) { return 64 + 256 * l; }   // h0 .. h7
constexpr int H_REMAP = 64 + 2048, H_F = H_REMAP + 256, H_DIR = H_F + 128, H_COLS = H_DIR + 32;   // 2528 halves per plane
constexpr int z_layer(int l) { return 256 * l; }         // dz0 .. dz7
constexpr int Z_REMAP = 2048, Z_F = 2304, Z_HEADS = 2432, Z_COLS = 2448;
constexpr int kGateLayers = 10;
static_assert(H_COLS % 8 == 0 && Z_COLS % 4 == 0, "16-byte rows");
// element offset of sample `row` of the segment that starts at column `col0` and is `width` wide
__host__ __device__ inline size_t seg_at(int col0, int width, long long m_pad, long long row) {
    return (size_t)col0 * (size_t)m_pad + (size_t)row * (size_t)width;
}

// ---- backward (input-gradient) chain: dH_in^T[in x samples] = W^T[in x out] * dZ^T[out x samples], layer after layer from
// the heads to layer 1.  k-steps of a layer: the gated gradient of the layer above as produced by the chain (SEG_ACT order),
// plus -- for the two layers fed by a head -- one k-step in natural order (vec32) carrying [d sigma, dz_r, dz_g, dz_b, 0...].
//   D0: rgb_layers.1^T   out 128 (f)        k = [heads(1)]               W = rgb1[c-1][row],  c = 1..3
//   D1: rgb_layers.0^T   out 256 (remap)    k = [dz_f(4)]                W = rgb0[col][row],  row < 256
//   D2: remap^T|sigma^T  out 256 (h7)       k = [dz_remap(8) | heads(1)] W = remap[col][row] | sigma[0][row] at c = 0
//   D3..D9: base_layers[7..1]^T  out 256    k = [dz_l(8)]                W = W_l[col][row (+63 for l = 5: cat(pe, h))]
constexpr int kDgradLayers = 10;
constexpr int kDgradKS[kDgradLayers] = {1, 4, 9, 8, 8, 8, 8, 8, 8, 8};
constexpr int kDgradRT[kDgradLayers] = {8, 16, 16, 16, 16, 16, 16, 16, 16, 16};
constexpr int dgrad_frag0(int d) {
    int f = 0;
    for (int i = 0; i < d; ++i) f += kDgradKS[i] * kDgradRT[i];
    return f;
}
constexpr int kDgradFrags = dgrad_frag0(kDgradLayers);   // 8 + 64 + 144 + 7 * 128 = 1112
static_assert(kDgradFrags == 1112, "dgrad fragment count");

// ---- weight-gradient jobs: dW[n, k] = sum_m dZ[m, n] * Hin[m, k].  A job is one linear, or one 128-row half of a 256-row
// linear (a workgroup keeps the job's whole tile set in accumulators: 8 waves x 1 row tile x up to 20 column tiles).
struct WgradJob {
    int layer;          // index into the 12 linears (0..7 base, 8 sigma, 9 remap, 10 rgb0, 11 rgb1)
    int z_col, n;       // dZ stash segment of the layer, rows of dW this job covers (multiple of 16, at most 128)
    int n0, z_n;        // first fragment-order row of the segment the job covers; width of the segment
    int k_col[2], k_n[2];   // up to two H stash segments (whole, fragment order) that make up the layer's input
};
constexpr int kWgradJobs = 21;
#define TGTC_WG_HALVES(layer, zc, kc0, kc1, kn0, kn1) {layer, zc, 128, 0, 256, {kc0, kc1}, {kn0, kn1}}, {layer, zc, 128, 128, 256, {kc0, kc1}, {kn0, kn1}}
constexpr WgradJob kWgradJob[kWgradJobs] = {
    TGTC_WG_HALVES(0, z_layer(0), H_PE, 0, 64, 0),
    TGTC_WG_HALVES(1, z_layer(1), h_layer(0), 0, 256, 0),
    TGTC_WG_HALVES(2, z_layer(2), h_layer(1), 0, 256, 0),
    TGTC_WG_HALVES(3, z_layer(3), h_layer(2), 0, 256, 0),
    TGTC_WG_HALVES(4, z_layer(4), h_layer(3), 0, 256, 0),
    TGTC_WG_HALVES(5, z_layer(5), h_layer(4), H_PE, 256, 64),     // cat(pe, h): fragment order here is [h | pe]
    TGTC_WG_HALVES(6, z_layer(6), h_layer(5), 0, 256, 0),
    TGTC_WG_HALVES(7, z_layer(7), h_layer(6), 0, 256, 0),
    {8, Z_HEADS, 16, 0, 16, {h_layer(7), 0}, {256, 0}},            // sigma_layer: row 0 of the heads tile
    TGTC_WG_HALVES(9, Z_REMAP, h_layer(7), 0, 256, 0),
    {10, Z_F, 128, 0, 128, {H_REMAP, H_DIR}, {256, 32}},
    {11, Z_HEADS, 16, 0, 16, {H_F, 0}, {128, 0}},                  // rgb_layers.1: rows 1..3 of the heads tile
};
#undef TGTC_WG_HALVES

}  // namespace train
}  // namespace tgtc
