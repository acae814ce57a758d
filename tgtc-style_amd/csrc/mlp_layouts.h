// Compile-time layouts of the packed networks (fragment / bias offsets per layer).  Shared by the
// NeRF kernels (mlp_nerf.hip) and the stylised kernels (mlp_style.hip); the packers check them at
// handle creation.
#pragma once
#include "mlp_core.h"

namespace tgtc {

// Stream order and compile-time fragment / bias bookkeeping (must match nerf_specs() below).
//   layer      L0  L1  L2  L3  L4  L5  L6  L7  SIG REMAP C0  C1
//   k-steps     2   8   8   8   8  10   8   8   8    8    9   4
//   row tiles  16  16  16  16  16  16  16  16   1   16    8   1
constexpr int kNerfKS[12] = {2, 8, 8, 8, 8, 10, 8, 8, 8, 8, 9, 4};
constexpr int kNerfRT[12] = {16, 16, 16, 16, 16, 16, 16, 16, 1, 16, 8, 1};
constexpr int nerf_frag0(int l) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += kNerfKS[i] * kNerfRT[i];
    return f;
}
constexpr int nerf_bias0(int l) {
    int b = 0;
    for (int i = 0; i < l; ++i) b += 16 * kNerfRT[i];
    return b;
}
struct NerfLayout {
    static constexpr int frag0(int l) { return nerf_frag0(l); }
    static constexpr int bias0(int l) { return nerf_bias0(l); }
    static constexpr int kFragsSigma = nerf_frag0(9);   // trunk + sigma head
    static constexpr int kFragsFull = nerf_frag0(12);   // 1172
    static constexpr int kBiasFloats = nerf_bias0(12);  // 2464
};

constexpr int kNerfBiasBytes = 16384;  // 2464 floats padded to a multiple of 8 KiB (up to 8 waves x 1 KiB LDS-DMA)
static_assert(NerfLayout::kBiasFloats * 4 <= kNerfBiasBytes, "bias table");
static_assert(NerfLayout::kFragsFull == 1172, "fragment count");


// ---- StyleMLP_before_concat (reference models.py:120-147): 5 linears 95,288,288,288,351 -> 256
//   k-steps: L0 [pe(2) | z(1)]   L1-3 [h(8) | z(1)]   L4 [h(8) | z(1) | pe(2)]
constexpr int kConcatKS[5] = {3, 9, 9, 9, 11};
constexpr int concat_frag0(int l) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += kConcatKS[i] * 16;
    return f;
}
constexpr int kConcatFrags = concat_frag0(5);  // 656
constexpr int kConcatBiasFloats = 5 * 256;

// ---- StyleMLP_Wild_multilayers (reference models.py:149-180): 8 linears 607,288,288,288,351,288,288 -> 256; 288 -> 3
//   k-steps: L0 [remap(8) | cf(8) | pe(2) | z(1)]   L1-3,5,6 [h(8) | z(1)]   L4 [h(8) | z(1) | pe(2)]   L7 [h(8) | z(1)]
constexpr int kStyleKS[8] = {19, 9, 9, 9, 11, 9, 9, 9};
constexpr int kStyleRT[8] = {16, 16, 16, 16, 16, 16, 16, 1};
constexpr int style_frag0(int l) {
    int f = 0;
    for (int i = 0; i < l; ++i) f += kStyleKS[i] * kStyleRT[i];
    return f;
}
constexpr int style_bias0(int l) {
    int b = 0;
    for (int i = 0; i < l; ++i) b += 16 * kStyleRT[i];
    return b;
}
constexpr int kStyleFrags = style_frag0(8);       // 1209
constexpr int kStyleBiasFloats = style_bias0(8);  // 1808
static_assert(kConcatFrags == 656 && kStyleFrags == 1209, "fragment counts");

// bias table of a style pair in LDS: [concat 1280 floats][style 1808 floats], padded to 4 KiB multiples
constexpr int kStylePairBiasBytes = 16384;
static_assert((kConcatBiasFloats + kStyleBiasFloats) * 4 <= kStylePairBiasBytes, "style bias table");

}  // namespace tgtc
