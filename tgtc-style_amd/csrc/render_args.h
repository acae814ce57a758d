// Launch arguments and host entry points of the fused ray kernels (render_fused.hip, render_styled_fused.hip), shared
// with the host chains in render.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace tgtc {

struct FusedArgs {
    const double* rays_o;
    const double* rays_d;
    long long R;
    int NC, NF;
    float near_, far_;
    const float* jitter;  // [R, NC] stratified-jitter uniforms, or null (utils.py:518-524)
    const char* net_c;    // coarse handle: bias region (kNerfBiasBytes) followed by the packed stream
    const char* net_f;    // fine handle
    float* rgb;           // [R, 3]
    float* t;             // [R]
    float* ts_out;        // depths-only kernel (PF = kFusedDepthsOnly): [R, NC + NF] merged fine-pass depths, ascending
};

struct FusedStyledArgs {
    FusedArgs ray;              // rays, sample counts, jitter, coarse / fine NeRF handles, pixel outputs
    const float* z;             // [R, 32] per-ray latent (rendering.py:118-127)
    const char* pair_bias;      // style handle: concat | style bias table (kStylePairBiasBytes)
    const char* concat_stream;
    const char* style_stream;
    char* slab;                 // n_wg * kStashBytesPerWG
};

int launch_fused_render(int prec_c, int prec_f, const FusedArgs& a, hipStream_t st);
bool fused_render_supports(int prec_c, int prec_f, int n_coarse, int n_fine);
int launch_fused_depths(int prec_c, const FusedArgs& a, hipStream_t st);
bool fused_depths_supports(int prec_c, int n_coarse, int n_fine);
int launch_fused_styled(int prec_c, const FusedStyledArgs& a, int n_wg, hipStream_t st);
bool fused_styled_supports(int prec_c, int prec_f, int prec_style, int n_coarse, int n_fine);

}  // namespace tgtc
