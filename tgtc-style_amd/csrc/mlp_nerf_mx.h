// Interface between mlp_nerf.hip (C ABI, dispatch, timing hook) and mlp_nerf_mx.hip (the fp16+fp6 kernels).
#pragma once
#include <vector>

#include "mlp_nerf_front.h"
#include "mlp_pack.h"

namespace tgtc {

std::vector<LayerSpec> nerf_specs(const tgtc_linear* l);  // mlp_nerf.hip

// packs the 12 linears for TGTC_PREC_FP16_FP6: `bias` = the kNerfBiasBytes region (fp32 biases + row exponents),
// `stream` = the group stream (mlp_mx.h)
int nerf_mx_pack(const tgtc_linear* layers, std::vector<char>& bias, std::vector<char>& stream);
// launches the kernel for (in_mode, full); no event handling
int nerf_mx_launch(int in_mode, bool full, const NerfArgs& a, hipStream_t st);
// the FULL network without the base_remap output, or densities over rays: two column tiles per wave, one wave per SIMD, persistent
// (mlp_nerf_mx2.hip)
int nerf_mx2_launch(int in_mode, bool full, const NerfArgs& a, hipStream_t st);
// fp16x3, rays in, sigma out (the coarse pass): two column tiles per wave, one wave per SIMD, persistent (mlp_nerf_x3s.hip)
int nerf_x3s_launch(const NerfArgs& a, hipStream_t st);

}  // namespace tgtc
