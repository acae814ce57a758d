// Fused render chains: the reference's per-batch sequence of callables collapsed into one host call
// that enqueues the kernels back to back on one stream (no host sync, no allocation).
//   plain : rendering.py:27-51   (cal_geometry)
//   styled: rendering.py:118-178 (render_style)   -- see mlp_style.hip
#include "common.h"

#include "mlp_pack.h"
#include "render_args.h"

namespace tgtc {
int launch_composite(const float* rgb, const float* sigma, const float* ts, int64_t R, int N, float* rgb_exp,
                     float* t_exp, float* weights, hipStream_t st);
int launch_sample_fine(const double* rays_o, const double* rays_d, const float* ts, const float* weights, int64_t R,
                       int N, int n_fine, double* pts_out, float* ts_out, hipStream_t st);
int nerf_forward_rays_impl(const tgtc_net* net, const double* rays_o, const double* rays_d, const float* ts, int64_t R,
                           int N, float* rgb, float* sigma, hipStream_t st);

int styled_forward_rays_impl(const tgtc_net* nerf, const tgtc_net* style, const double* rays_o, const double* rays_d,
                             const float* ts, const float* z, int64_t R, int N, float* rgb, float* sigma,
                             hipStream_t st);

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RenderWorkspace {
    float *ts_c, *sigma_c, *rgb_c, *w_c, *ts_f, *sigma_f, *rgb_f;
    size_t total;
    RenderWorkspace(char* base, int64_t R, int nc, int nf) {
        const int nt = nc + nf;
        size_t off = 0;
        auto take = [&](size_t floats) {
            float* p = reinterpret_cast<float*>(base + off);
            off += align256(floats * sizeof(float));
            return p;
        };
        ts_c = take((size_t)R * nc);
        sigma_c = take((size_t)R * nc);
        rgb_c = take((size_t)R * nc * 3);
        w_c = take((size_t)R * nc);
        ts_f = take((size_t)R * nt);
        sigma_f = take((size_t)R * nt);
        rgb_f = take((size_t)R * nt * 3);
        total = off;
    }
};
}  // namespace tgtc

using namespace tgtc;

extern "C" size_t tgtc_render_workspace_bytes(int64_t R, int n_coarse, int n_fine) {
    if (R < 0 || n_coarse < 0 || n_fine < 0) return 0;
    return RenderWorkspace(nullptr, R, n_coarse, n_fine).total;
}

// The plain render.  Whenever the sample counts and precisions allow it (and the caller does not ask for the coarse
// image) this is ONE launch of the fused ray kernel (render_fused.hip) and the workspace is not touched; otherwise
// the chain of per-sample kernels below runs (tgtc_render_rays_plain_chain, always available).
extern "C" int tgtc_render_rays_plain_fused(const tgtc_net* coarse, const tgtc_net* fine, const double* rays_o,
                                            const double* rays_d, int64_t R, int n_coarse, int n_fine, float near_,
                                            float far_, const float* jitter, float* rgb_fine, float* t_fine, void* stream) {
    TGTC_REQUIRE(coarse && fine && R >= 0, "render_rays_plain_fused: bad argument");
    if (!(coarse->kind == 0 && fine->kind == 0 && fused_render_supports(coarse->precision, fine->precision, n_coarse, n_fine)))
        return fail(TGTC_ERR_UNSUPPORTED, "render_rays_plain_fused: no single-kernel build for precisions %d + %d with %d + %d samples",
                    coarse->precision, fine->precision, n_coarse, n_fine);
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rays_o && rays_d && rgb_fine && t_fine, "render_rays_plain_fused: null pointer");
    FusedArgs a{rays_o, rays_d, R, n_coarse, n_fine, near_, far_, jitter, coarse->dev, fine->dev, rgb_fine, t_fine, nullptr};
    return launch_fused_render(coarse->precision, fine->precision, a, as_stream(stream));
}

extern "C" int tgtc_render_rays_plain(const tgtc_net* coarse, const tgtc_net* fine, const double* rays_o,
                                      const double* rays_d, int64_t R, int n_coarse, int n_fine, float near_,
                                      float far_, const float* jitter, void* workspace, size_t workspace_bytes,
                                      float* rgb_fine, float* t_fine, float* rgb_coarse, float* t_coarse,
                                      void* stream) {
    TGTC_REQUIRE(coarse && fine && R >= 0, "render_rays_plain: bad argument");
    if (!rgb_coarse && !t_coarse && coarse->kind == 0 && fine->kind == 0 &&
        fused_render_supports(coarse->precision, fine->precision, n_coarse, n_fine)) {
        // fp16x3 + fp16_fp6: the fine pass is faster on the two-tile per-sample kernel (mlp_nerf_mx2.hip: half the LDS bytes per
        // MFMA of any one-tile loop, the ray kernel's included) than inside the ray kernel, and the per-sample tensors it
        // needs are 0.3 % of the frame time in HBM traffic: a caller that hands over the workspace gets the split path
        const bool split_is_faster = coarse->precision == TGTC_PREC_FP16X3 && fine->precision == TGTC_PREC_FP16_FP6 && workspace &&
                                     workspace_bytes >= tgtc_render_workspace_bytes(R, n_coarse, n_fine);
        if (!split_is_faster)
            return tgtc_render_rays_plain_fused(coarse, fine, rays_o, rays_d, R, n_coarse, n_fine, near_, far_, jitter, rgb_fine, t_fine, stream);
    }
    return tgtc_render_rays_plain_chain(coarse, fine, rays_o, rays_d, R, n_coarse, n_fine, near_, far_, jitter, workspace,
                                        workspace_bytes, rgb_fine, t_fine, rgb_coarse, t_coarse, stream);
}

extern "C" int tgtc_render_rays_plain_chain(const tgtc_net* coarse, const tgtc_net* fine, const double* rays_o,
                                            const double* rays_d, int64_t R, int n_coarse, int n_fine, float near_,
                                            float far_, const float* jitter, void* workspace, size_t workspace_bytes,
                                            float* rgb_fine, float* t_fine, float* rgb_coarse, float* t_coarse,
                                            void* stream) {
    TGTC_REQUIRE(coarse && fine && R >= 0, "render_rays_plain: bad argument");
    // the reference dereferences None when N_samples_fine == 0; require it instead
    TGTC_REQUIRE(n_coarse >= 3 && n_fine >= 1, "render_rays_plain: need n_coarse >= 3 and n_fine >= 1 (got %d, %d)",
                 n_coarse, n_fine);
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rays_o && rays_d && workspace && rgb_fine && t_fine, "render_rays_plain: null pointer");
    RenderWorkspace ws(static_cast<char*>(workspace), R, n_coarse, n_fine);
    TGTC_REQUIRE(workspace_bytes >= ws.total, "render_rays_plain: workspace of %zu bytes, need %zu", workspace_bytes,
                 ws.total);
    hipStream_t st = as_stream(stream);
    int rc = tgtc_sample_coarse(rays_o, rays_d, R, n_coarse, near_, far_, jitter, nullptr, ws.ts_c, stream);
    if (rc) return rc;
    // coarse pass: only the weights are consumed unless the caller asks for the coarse image
    float* rgb_c = rgb_coarse ? ws.rgb_c : nullptr;
    rc = nerf_forward_rays_impl(coarse, rays_o, rays_d, ws.ts_c, R, n_coarse, rgb_c, ws.sigma_c, st);
    if (rc) return rc;
    rc = launch_composite(rgb_c, ws.sigma_c, ws.ts_c, R, n_coarse, rgb_coarse, t_coarse, ws.w_c, st);
    if (rc) return rc;
    rc = launch_sample_fine(rays_o, rays_d, ws.ts_c, ws.w_c, R, n_coarse, n_fine, nullptr, ws.ts_f, st);
    if (rc) return rc;
    rc = nerf_forward_rays_impl(fine, rays_o, rays_d, ws.ts_f, R, n_coarse + n_fine, ws.rgb_f, ws.sigma_f, st);
    if (rc) return rc;
    return launch_composite(ws.rgb_f, ws.sigma_f, ws.ts_f, R, n_coarse + n_fine, rgb_fine, t_fine, nullptr, st);
}

// rendering.py:118-178 (render_style).  Whenever the sample counts and precisions allow it (fp16x3 everywhere) and the caller
// does not ask for the coarse image this is ONE launch of the stylised ray kernel (render_styled_fused.hip) and the workspace is
// not touched; otherwise the chain of per-sample kernels below runs (tgtc_render_rays_styled_chain, always available).
extern "C" int tgtc_render_rays_styled(const tgtc_net* coarse, const tgtc_net* fine, const tgtc_net* style,
                                       const double* rays_o, const double* rays_d, const float* z, int64_t R,
                                       int n_coarse, int n_fine, float near_, float far_, const float* jitter,
                                       void* workspace, size_t workspace_bytes, float* rgb_fine, float* t_fine,
                                       float* rgb_coarse, float* t_coarse, void* stream) {
    TGTC_REQUIRE(coarse && fine && style && R >= 0, "render_rays_styled: bad argument");
    if (!rgb_coarse && !t_coarse && coarse->kind == 0 && fine->kind == 0 && style->kind == 1 &&
        fused_styled_supports(coarse->precision, fine->precision, style->precision, n_coarse, n_fine)) {
        if (R == 0) return TGTC_OK;
        TGTC_REQUIRE(rays_o && rays_d && z && rgb_fine && t_fine, "render_rays_styled: null pointer");
        FusedStyledArgs a{};
        a.ray = FusedArgs{rays_o, rays_d, R, n_coarse, n_fine, near_, far_, jitter, coarse->dev, fine->dev, rgb_fine, t_fine, nullptr};
        a.z = z, a.pair_bias = style->dev, a.concat_stream = style->dev + style->bias_bytes;
        a.style_stream = style->dev + style->stream2_off, a.slab = style->dev + style->stash_off;
        return launch_fused_styled(coarse->precision, a, style->n_wg, as_stream(stream));
    }
    return tgtc_render_rays_styled_chain(coarse, fine, style, rays_o, rays_d, z, R, n_coarse, n_fine, near_, far_, jitter, workspace,
                                         workspace_bytes, rgb_fine, t_fine, rgb_coarse, t_coarse, stream);
}

// The stylised chain of per-sample kernels: like the plain chain, with the stylised colour on both passes.
// The coarse colours only matter if the caller asks for the coarse image: the fine sampler consumes the
// weights, which depend on sigma alone, so by default the coarse pass runs the sigma-only NeRF kernel.
extern "C" int tgtc_render_rays_styled_chain(const tgtc_net* coarse, const tgtc_net* fine, const tgtc_net* style,
                                             const double* rays_o, const double* rays_d, const float* z, int64_t R,
                                             int n_coarse, int n_fine, float near_, float far_, const float* jitter,
                                             void* workspace, size_t workspace_bytes, float* rgb_fine, float* t_fine,
                                             float* rgb_coarse, float* t_coarse, void* stream) {
    TGTC_REQUIRE(coarse && fine && style && R >= 0, "render_rays_styled: bad argument");
    TGTC_REQUIRE(n_coarse >= 3 && n_fine >= 1, "render_rays_styled: need n_coarse >= 3 and n_fine >= 1 (got %d, %d)",
                 n_coarse, n_fine);
    if (R == 0) return TGTC_OK;
    TGTC_REQUIRE(rays_o && rays_d && z && workspace && rgb_fine && t_fine, "render_rays_styled: null pointer");
    RenderWorkspace ws(static_cast<char*>(workspace), R, n_coarse, n_fine);
    TGTC_REQUIRE(workspace_bytes >= ws.total, "render_rays_styled: workspace of %zu bytes, need %zu", workspace_bytes,
                 ws.total);
    hipStream_t st = as_stream(stream);
    int rc;
    if (!rgb_coarse && !t_coarse && coarse->kind == 0 && fused_depths_supports(coarse->precision, n_coarse, n_fine)) {
        // no coarse image wanted: coarse depths -> coarse sigma -> weights -> fine depths never leave the ray kernel
        FusedArgs a{rays_o, rays_d, R, n_coarse, n_fine, near_, far_, jitter, coarse->dev, coarse->dev, nullptr, nullptr, ws.ts_f};
        rc = launch_fused_depths(coarse->precision, a, st);
        if (rc) return rc;
        rc = styled_forward_rays_impl(fine, style, rays_o, rays_d, ws.ts_f, z, R, n_coarse + n_fine, ws.rgb_f, ws.sigma_f, st);
        if (rc) return rc;
        return launch_composite(ws.rgb_f, ws.sigma_f, ws.ts_f, R, n_coarse + n_fine, rgb_fine, t_fine, nullptr, st);
    }
    rc = tgtc_sample_coarse(rays_o, rays_d, R, n_coarse, near_, far_, jitter, nullptr, ws.ts_c, stream);
    if (rc) return rc;
    float* rgb_c = rgb_coarse ? ws.rgb_c : nullptr;
    if (rgb_c)
        rc = styled_forward_rays_impl(coarse, style, rays_o, rays_d, ws.ts_c, z, R, n_coarse, rgb_c, ws.sigma_c, st);
    else
        rc = nerf_forward_rays_impl(coarse, rays_o, rays_d, ws.ts_c, R, n_coarse, nullptr, ws.sigma_c, st);
    if (rc) return rc;
    rc = launch_composite(rgb_c, ws.sigma_c, ws.ts_c, R, n_coarse, rgb_coarse, t_coarse, ws.w_c, st);
    if (rc) return rc;
    rc = launch_sample_fine(rays_o, rays_d, ws.ts_c, ws.w_c, R, n_coarse, n_fine, nullptr, ws.ts_f, st);
    if (rc) return rc;
    rc = styled_forward_rays_impl(fine, style, rays_o, rays_d, ws.ts_f, z, R, n_coarse + n_fine, ws.rgb_f, ws.sigma_f, st);
    if (rc) return rc;
    return launch_composite(ws.rgb_f, ws.sigma_f, ws.ts_f, R, n_coarse + n_fine, rgb_fine, t_fine, nullptr, st);
}
