/*
 * tgtc_hip.h -- C ABI of the MI355X (gfx950) render hot path of TGTC-Style.
 *
 * One shared library (libtgtc_hip.so) of hand-written HIP kernels.  Every entry point is
 * `extern "C"`, takes plain device pointers + sizes + a hipStream_t (passed as void*), writes only
 * into caller-allocated outputs, never synchronises the device and never allocates inside a launch
 * function (network handles own their packed weights; they are created / destroyed explicitly).
 *
 * Return value: 0 = ok, <0 = error class (below); tgtc_last_error() gives the thread-local text.
 *
 * The reference (PaiDii/TGTC-Style) is pure Python on PyTorch; there is no FFI in it.  Each entry
 * point replaces the reference *Python callable* cited beside it (file:line under /root/reference).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Layouts (all row-major, dense):
 *   rays_o, rays_d   double [R,3]        (the reference keeps rays in float64: dataset.py:420-429)
 *   ts               float  [R,N]        sample depths
 *   pts              double [R,N,3]
 *   rgb              float  [R,N,3]      sigma float [R,N]
 *   per-ray outputs  float  [R,3] / [R]
 */
#ifndef TGTC_HIP_H
#define TGTC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGTC_OK 0
#define TGTC_ERR_ARG (-1)         /* null pointer, negative size, bad enum */
#define TGTC_ERR_UNSUPPORTED (-2) /* a shape / configuration the kernels are not built for */
#define TGTC_ERR_HIP (-3)         /* a HIP runtime call failed */

/* Arithmetic mode of the fused MLP kernels (MFMA operands; accumulation is always fp32). */
#define TGTC_PREC_FP16X3 0 /* split fp16 (hi+lo) x 3 MFMA products: fp32-equivalent, the parity mode */
#define TGTC_PREC_FP16 1   /* single fp16 MFMA product: fastest, ~2e-3 abs error on composited RGB */
#define TGTC_PREC_FP16_FP6 2 /* fp16 product + two block-scaled fp6 (e2m3) correction products: ~1e-4, NeRF nets only */

typedef struct tgtc_net tgtc_net; /* opaque: packed weights of one network, resident in HBM */

/* One nn.Linear in the reference layout: weight [out_features, in_features] row-major, bias [out]. HOST pointers. */
typedef struct {
    const float* weight;
    const float* bias;
    int32_t out_features;
    int32_t in_features;
} tgtc_linear;

int tgtc_version(void);
const char* tgtc_last_error(void);

/* ------------------------------------------------------------------ a1+a2: ray generation
 * dataset.py:33-42 (get_rays_np) + dataset.py:44-61 (ndc_rays_np, near given by caller; the datasets pass 1.0).
 * Generates rays for pixels [first_pixel, first_pixel+n) of an H x W frame in row-major pixel order
 * (so ranks can generate their own shard).  c2w: 12 HOST floats (3x4 row-major).  ndc=0 skips the warp. */
int tgtc_gen_rays(int H, int W, double fx, double fy, double cx, double cy, const float* c2w,
                  int pixel_alignment, int ndc, double ndc_near, int64_t first_pixel, int64_t n,
                  double* rays_o, double* rays_d, void* stream);

/* ------------------------------------------------------------------ a3: coarse sampling
 * utils.py:509-531 sampling_pts_uniform (harmony=False).  jitter: float [R,N] uniform(0,1) or NULL (perturb=False).
 * pts may be NULL (the fused path never materialises it). */
int tgtc_sample_coarse(const double* rays_o, const double* rays_d, int64_t R, int N, float near_, float far_,
                       const float* jitter, double* pts, float* ts, void* stream);

/* ------------------------------------------------------------------ a4: positional encoding
 * models.py:46-60 Embedder.forward with log-sampled bands 2^0..2^(L-1), include_input.  x: [M,3] double
 * (x_is_f64=1) or float; out: float [M, 3+6L] (the float32 cast of models.py:219-220 is applied). */
int tgtc_posenc(const void* x, int x_is_f64, int64_t M, int L, float* out, void* stream);

/* ------------------------------------------------------------------ a5: NeRF MLP
 * models.py:63-117 MLP_style / :182-223 StyleNerf with D=8, W=256, skips=[4], use_viewdir, ReLU, PE 10/4.
 * layers: the 12 linears in the order of MLP_style.layers (models.py:93):
 *   base_layers[0..7], sigma_layer, base_remap_layer, rgb_layers[0], rgb_layers[1]. */
int tgtc_nerf_create(const tgtc_linear* layers, int n_layers, int precision, tgtc_net** out);
int tgtc_net_destroy(tgtc_net* net);
int tgtc_net_precision(const tgtc_net* net);

/* StyleNerf.forward (models.py:216-223): raw points / view dirs [M,3] double -> outputs.  Any output may be NULL.
 * base_remap float [M,256]; pts_enc float [M,63]; dirs_enc float [M,27]. */
int tgtc_nerf_forward(const tgtc_net* net, const double* pts, const double* dirs, int64_t M,
                      float* rgb, float* sigma, float* base_remap, float* pts_enc, float* dirs_enc, void* stream);

/* MLP_style.forward (models.py:95-117) on already-encoded float inputs [M,63] / [M,27]. */
int tgtc_nerf_mlp_forward(const tgtc_net* net, const float* pts_enc, const float* dirs_enc, int64_t M,
                          float* rgb, float* sigma, float* base_remap, void* stream);

/* Hot path: points are never materialised; sample (r,i) is rays_o[r] + ts[r,i]*rays_d[r], dirs = rays_d[r]
 * (rendering.py:27-31).  need_rgb=0 skips the colour head (the coarse pass of a render only consumes weights
 * when its RGB is discarded; the reference still computes it). */
int tgtc_nerf_forward_rays(const tgtc_net* net, const double* rays_o, const double* rays_d, const float* ts,
                           int64_t R, int N, float* rgb, float* sigma, void* stream);

/* ------------------------------------------------------------------ a6: alpha compositing
 * utils.py:354-386 alpha_composition with sigma_noise_std=0, white_bkgd=False.  weights may be NULL. */
int tgtc_composite(const float* rgb, const float* sigma, const float* ts, int64_t R, int N,
                   float* rgb_exp, float* t_exp, float* weights, void* stream);

/* Training-side forms of a6 (train_tgtcs.py:218-309 differentiates through alpha_composition):
 * composite_train: sigma + noise in front of the ReLUs (noise float [R,N] = randn * sigma_noise_std drawn by the caller,
 *   utils.py:371-376, or NULL) and the optional white background (utils.py:381-384).
 * composite_backward: dL/d rgb [R,N,3] and dL/d sigma [R,N] from dL/d rgb_exp [R,3], dL/d t_exp [R], dL/d weights [R,N]
 *   (any of the three may be NULL = zero); either output may be NULL. */
int tgtc_composite_train(const float* rgb, const float* sigma, const float* ts, const float* noise, int white_bkgd,
                         int64_t R, int N, float* rgb_exp, float* t_exp, float* weights, void* stream);
int tgtc_composite_backward(const float* rgb, const float* sigma, const float* ts, const float* noise, int white_bkgd,
                            int64_t R, int N, const float* grad_rgb_exp, const float* grad_t_exp,
                            const float* grad_weights, float* grad_rgb, float* grad_sigma, void* stream);

/* ------------------------------------------------------------------ a7: fine sampling
 * utils.py:573-580 sampling_pts_fine_torch -> utils.py:583-609 sample_pdf(det=True), then the sorted merge.
 * ts [R,N], weights [R,N] -> ts_out [R,N+n_fine] ascending; pts_out [R,N+n_fine,3] double or NULL. */
int tgtc_sample_fine(const double* rays_o, const double* rays_d, const float* ts, const float* weights,
                     int64_t R, int N, int n_fine, double* pts_out, float* ts_out, void* stream);

/* ------------------------------------------------------------------ fused plain render (cal_geometry chain)
 * rendering.py:27-51: coarse sample -> NeRF(coarse) -> composite -> fine sample -> NeRF(fine) -> composite.
 * jitter: float [R,n_coarse] or NULL.  Outputs: rgb float [R,3], depth float [R]; optional coarse outputs.
 *
 * tgtc_render_rays_plain runs the whole chain as ONE persistent kernel (a wavefront owns a ray; per-sample
 * tensors never exist; HBM sees 48 B of ray in and 16 B of pixel out) whenever
 *   - n_coarse and n_coarse+n_fine are multiples of 16 (32 in TGTC_PREC_FP16), n_coarse <= 192, total <= 256,
 *   - the precisions are fp16x3+fp16x3, fp16x3 (coarse) + fp16_fp6 (fine), or fp16+fp16,
 *   - the coarse image is not requested (rgb_coarse == t_coarse == NULL);
 * then `workspace` is not touched and may be NULL.  Otherwise it falls back to
 * tgtc_render_rays_plain_chain: the same arithmetic as a sequence of per-sample kernels through
 * workspace (device scratch of at least tgtc_render_workspace_bytes(R, n_coarse, n_fine) bytes).
 * One pair takes the chain by choice: fp16x3 (coarse) + fp16_fp6 (fine) with a sufficient workspace handed over -- its fine
 * pass runs ~10 % faster on the two-tile per-sample kernel (csrc/mlp_nerf_mx2.hip) than inside the ray kernel, and the
 * per-sample tensors cost 0.3 % of the frame in HBM traffic.  tgtc_render_rays_plain_fused is the single kernel and
 * nothing else (TGTC_ERR_UNSUPPORTED outside the conditions above). */
size_t tgtc_render_workspace_bytes(int64_t R, int n_coarse, int n_fine);
int tgtc_render_rays_plain(const tgtc_net* coarse, const tgtc_net* fine, const double* rays_o, const double* rays_d,
                           int64_t R, int n_coarse, int n_fine, float near_, float far_, const float* jitter,
                           void* workspace, size_t workspace_bytes,
                           float* rgb_fine, float* t_fine, float* rgb_coarse, float* t_coarse, void* stream);
int tgtc_render_rays_plain_chain(const tgtc_net* coarse, const tgtc_net* fine, const double* rays_o,
                                 const double* rays_d, int64_t R, int n_coarse, int n_fine, float near_, float far_,
                                 const float* jitter, void* workspace, size_t workspace_bytes, float* rgb_fine,
                                 float* t_fine, float* rgb_coarse, float* t_coarse, void* stream);
int tgtc_render_rays_plain_fused(const tgtc_net* coarse, const tgtc_net* fine, const double* rays_o,
                                 const double* rays_d, int64_t R, int n_coarse, int n_fine, float near_, float far_,
                                 const float* jitter, float* rgb_fine, float* t_fine, void* stream);

/* ------------------------------------------------------------------ a8: latent table
 * models.py:490-506 StyleLatents_variational.forward.  latents float [S,F,D] device, mu float [S,D] device,
 * style_ids / frame_ids int64 [R] device.  tile7: the llff `repeat((7,1))` wrap (models.py:496). */
int tgtc_latents_forward(const float* latents, const float* mu, int S, int F, int D, const int64_t* style_ids,
                         const int64_t* frame_ids, int64_t R, float sigma_scale, int tile7, float* out, void* stream);
/* Its gradient for the training side (Style_train optimises the table, train_tgtcs.py:312-571): grad_out [R,D] ->
 * d_latents [S,F,D] += sigma_scale * g at the gathered rows, d_mu [S,D] += (1 - sigma_scale) * g; both buffers zeroed by the
 * caller, either may be NULL. */
int tgtc_latents_backward(const float* grad_out, int S, int F, int D, const int64_t* style_ids, const int64_t* frame_ids,
                          int64_t R, float sigma_scale, int tile7, float* d_latents, float* d_mu, void* stream);

/* ------------------------------------------------------------------ a13 (image epilogue)
 * rendering.py:66-71 (cal_geometry), :202-206 (render_style), :358-361 (render_train_style): what the drivers do to
 * every finished frame before imageio.imwrite -- per-frame sv_t = (t - min t) / (max t - min t + eps), then
 * np.array(x * 255, np.int32) and to8b = np.uint8 cast (utils.py:463; wraps modulo 256, e.g. 1.0039 -> 0).
 * rgb float [frames*pixels,3], t float [frames*pixels] -> rgb8 uint8 [frames*pixels,3], depth8 uint8 [frames*pixels]
 * (either output may be NULL).  eps = 1e-7 for cal_geometry / render_style, 0 for render_train_style (whose depth
 * image is the same plane written three times, host side).  Only 4 bytes per ray cross PCIe afterwards. */
int tgtc_image_epilogue(const float* rgb, const float* t, int64_t frames, int64_t pixels, float eps, unsigned char* rgb8,
                        unsigned char* depth8, void* stream);

/* ------------------------------------------------------------------ a10+a11: the two style MLPs
 * models.py:120-147 StyleMLP_before_concat (5 linears: 95,288,288,288,351 -> 256) and
 * models.py:149-180 StyleMLP_Wild_multilayers (8 linears: 607,288,288,288,351,288,288 -> 256, 288 -> 3). */
int tgtc_style_create(const tgtc_linear* concat_layers, int n_concat, const tgtc_linear* style_layers, int n_style,
                      int precision, tgtc_net** out);
/* x float [M,63], latent float [M,32] -> concat_features float [M,256] */
int tgtc_concat_mlp_forward(const tgtc_net* style, const float* x, const float* latent, int64_t M,
                            float* concat_features, void* stream);
/* x float [M,63], concated float [M,512], latent float [M,32] -> rgb float [M,3] */
int tgtc_style_mlp_forward(const tgtc_net* style, const float* x, const float* concated, const float* latent,
                           int64_t M, float* rgb, void* stream);
/* One stylised pass over rays (rendering.py:122-142): NeRF trunk (sigma, base_remap), concat MLP on the per-ray
 * latent z [R,32], style MLP on mean(z) broadcast.  Outputs rgb [R,N,3], sigma [R,N]. */
int tgtc_styled_forward_rays(const tgtc_net* nerf, const tgtc_net* style, const double* rays_o, const double* rays_d,
                             const float* ts, const float* z, int64_t R, int N, float* rgb, float* sigma, void* stream);
/* The stylised render of rays (rendering.py:109-182 render_style): like tgtc_render_rays_plain with the stylised colour
 * (per-ray latent z float [R,32]).  tgtc_render_rays_styled runs it as ONE persistent kernel (a wavefront owns a ray; coarse
 * passes, fine sampling, concat MLP + NeRF trunk + style MLP per fine tile and compositing back to back; no per-sample
 * tensor, `workspace` not touched and may be NULL) whenever the three handles are TGTC_PREC_FP16X3, the sample counts
 * satisfy the rule of tgtc_render_rays_plain and the coarse image is not requested; otherwise it falls back to
 * tgtc_render_rays_styled_chain, the same arithmetic as a sequence of per-sample kernels through `workspace`. */
int tgtc_render_rays_styled(const tgtc_net* coarse, const tgtc_net* fine, const tgtc_net* style,
                            const double* rays_o, const double* rays_d, const float* z, int64_t R, int n_coarse,
                            int n_fine, float near_, float far_, const float* jitter, void* workspace,
                            size_t workspace_bytes, float* rgb_fine, float* t_fine, float* rgb_coarse,
                            float* t_coarse, void* stream);
int tgtc_render_rays_styled_chain(const tgtc_net* coarse, const tgtc_net* fine, const tgtc_net* style,
                                  const double* rays_o, const double* rays_d, const float* z, int64_t R, int n_coarse,
                                  int n_fine, float near_, float far_, const float* jitter, void* workspace,
                                  size_t workspace_bytes, float* rgb_fine, float* t_fine, float* rgb_coarse,
                                  float* t_coarse, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TGTC_HIP_H */
