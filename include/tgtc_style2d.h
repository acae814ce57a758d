/*
 * tgtc_style2d.h -- C ABI of the 2-D style pass: patch embedding, the
 * style transformer (3 style-encoder + 3 content-encoder + 3 decoder layers, d=512, 8 heads, FFN 2048,
 * post-norm), the CNN decoder, the VGG-19 prefix up to relu4_1, calc_mean_std / AdaIN and the
 * post-processing of trans_test.py.  Forward only (eval mode: dropout is the identity).
 *
 * Every matrix product (linear layers, QK^T, PV, 8x8 / 3x3 / 1x1 convolutions as implicit GEMMs over
 * LDS-staged tiles) runs on v_mfma_f32_16x16x32_f16 with fp32 accumulation; `precision` selects split-fp16
 * (3 products, fp32-equivalent, default) or single fp16.
 *
 * Layouts.  Images and VGG / decoder outputs: float NCHW with N = 1 (the reference's), i.e. [C,H,W].
 * Token / feature maps inside the pass: token-major [h*w, C] ("NHWC"), which is what the transformer
 * flattens to (transformer.py:56-60); hs is returned token-major and `tgtc_s2d_tokens_to_nchw` converts.
 *
 * All pointers are device pointers unless noted; `workspace` is caller-allocated scratch of at least the
 * size the matching *_workspace_bytes function reports; nothing synchronises or allocates in a launch.
 */
#ifndef TGTC_STYLE2D_H
#define TGTC_STYLE2D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgtc_style2d tgtc_style2d; /* opaque: device copies of the 2-D networks' parameters */

/* One named parameter tensor (HOST pointer), names as in the reference state dicts:
 *   transformer.*  -> "encoder_c.layers.0.qk.weight", ... (transformer.py:13-44; 142 keys, new_ps.* ignored)
 *   embedding      -> "proj.weight" [512,3,8,8], "proj.bias"                       (tctrans.py:26)
 *   decoder        -> "1.weight", "1.bias", "5.weight", ... (Sequential indices)  (tctrans.py:36-66)
 *   vgg            -> "0.weight", "0.bias", "2.weight", ... up to "29.*"          (tctrans.py:68-99)
 * Any of the four groups may be absent (count 0); calling an op whose group is absent is an error. */
typedef struct {
    const char* name;
    const float* data;
    int64_t numel;
} tgtc_named_tensor;

int tgtc_s2d_create(const tgtc_named_tensor* transformer, int n_transformer, const tgtc_named_tensor* embedding,
                    int n_embedding, const tgtc_named_tensor* decoder, int n_decoder, const tgtc_named_tensor* vgg,
                    int n_vgg, int precision, tgtc_style2d** out);
int tgtc_s2d_destroy(tgtc_style2d* h);

/* a14  PatchEmbed.forward (tctrans.py:29-33): Conv2d(3,512,k=8,s=8).  img [3,H,W] -> tokens [(H/8)*(W/8), 512]. */
int tgtc_s2d_patch_embed(const tgtc_style2d* h, const float* img, int H, int W, float* tokens, void* stream);

/* a15-a17  Transformer.forward(style, None, content, pos_c=content, pos_s=None) (transformer.py:46-75).
 * style_tokens [Ns,512], content_tokens [Nc,512] (patch embeddings) -> hs [Nc,512] token-major. */
size_t tgtc_s2d_transformer_workspace_bytes(int n_style_tokens, int n_content_tokens);
int tgtc_s2d_transformer_forward(const tgtc_style2d* h, const float* style_tokens, int n_style_tokens,
                                 const float* content_tokens, int n_content_tokens, void* workspace,
                                 size_t workspace_bytes, float* hs, void* stream);

/* Granular seams used by the parity tests (same kernels the full forward enqueues).
 * mha: nn.MultiheadAttention forward of layer `prefix` (e.g. "decoder.layers.0.multihead_attn."),
 *      query [L,512], key/value [S,512] -> out [L,512].
 * encoder_layer: TransformerEncoderLayer.forward_post (transformer.py:167-184); has_pos selects the qk / qkv branch.
 * decoder_layer: TransformerDecoderLayer.forward_post with pos=None (transformer.py:236-263). */
int tgtc_s2d_mha(const tgtc_style2d* h, const char* prefix, const float* query, int L, const float* key,
                 const float* value, int S, void* workspace, size_t workspace_bytes, float* out, void* stream);
int tgtc_s2d_encoder_layer(const tgtc_style2d* h, const char* prefix, const float* src, int S, int has_pos,
                           void* workspace, size_t workspace_bytes, float* out, void* stream);
int tgtc_s2d_decoder_layer(const tgtc_style2d* h, const char* prefix, const float* tgt, int L, const float* memory,
                           int S, const float* query_pos, void* workspace, size_t workspace_bytes, float* out,
                           void* stream);

/* a18  CNN decoder (tctrans.py:36-66): tokens [h*w,512] -> image [3, 8h, 8w]. */
size_t tgtc_s2d_decode_workspace_bytes(int h, int w);
int tgtc_s2d_cnn_decode(const tgtc_style2d* hd, const float* tokens, int h, int w, void* workspace,
                        size_t workspace_bytes, float* image, void* stream);

/* a19  StyTrans.encode_with_intermediate over vgg[:31] (tctrans.py:161-166): img [3,H,W] -> relu1_1 [64,H,W],
 * relu2_1 [128,ceil(H/2),ceil(W/2)], relu3_1 [256,..], relu4_1 [512,..] (NCHW).  Any output may be NULL. */
size_t tgtc_s2d_vgg_workspace_bytes(int H, int W);
int tgtc_s2d_vgg_encode(const tgtc_style2d* h, const float* img, int H, int W, void* workspace,
                        size_t workspace_bytes, float* relu1_1, float* relu2_1, float* relu3_1, float* relu4_1,
                        void* stream);

/* a20  calc_mean_std (function.py:4-12 == Style_function.py:4-12): feat [C,HW] -> mean [C], std [C] = sqrt(unbiased var + eps). */
int tgtc_s2d_mean_std(const float* feat, int C, int64_t HW, float eps, float* mean, float* std_, void* stream);
/* adaptive_instance_normalization (Style_function.py:15-24) with eps = 1e-5: content, style [C,HWc]/[C,HWs] -> out [C,HWc].
 * stats: scratch of 4*C floats. */
int tgtc_s2d_adain(const float* content, int64_t HWc, const float* style, int64_t HWs, int C, float* stats, float* out,
                   void* stream);

/* A dense layer on device pointers: y[M,N] = act(x[M,K] . W[N,K]^T + b[N]) (nn.Linear layout; b may be NULL; relu 0/1;
 * precision TGTC_PREC_FP16 or TGTC_PREC_FP16X3).  For the few nn.Linear stacks of the reference that sit outside the
 * fused kernels: the VAE encoder that initialises the latent table when no latent checkpoint exists
 * (models.py:371-395 VAE_encoder, train_tgtcs.py:148-155). */
int tgtc_s2d_linear(const float* x, int64_t M, int K, const float* W, const float* b, int N, int relu, int precision,
                    float* y, void* stream);

/* The same with the weight also handed in as fp16 hi / lo halves (made by tgtc_s2d_split, n = N*K; used when K % 4 == 0): the
 * GEMM loads them without converting. */
int tgtc_s2d_split(const float* w, int64_t n, void* hi, void* lo, void* stream);
int tgtc_s2d_linear_pre(const float* x, int64_t M, int K, const float* W, const void* W_hi, const void* W_lo, const float* b,
                        int N, int relu, int precision, float* y, void* stream);

/* Backward of that layer for the training side (reference train_tgtcs.py:218-309 backpropagates through the NeRF MLPs):
 *   dx[M,K] = dy[M,N] . W[N,K]        (NULL to skip)
 *   dW[N,K] = dy^T . x,  db[N] = column sums of dy   (NULL to skip either)
 * as GEMMs on the same kernel: dy and x are transposed into the workspace, the sample dimension is split over GEMM batches
 * and the partial products summed.  relu_y (NULL, or the forward's output [M,N] when it fused a ReLU) gates dy on the fly:
 * elements with relu_y <= 0 carry no gradient. */
size_t tgtc_s2d_linear_backward_workspace_bytes(int64_t M, int K, int N);
int tgtc_s2d_linear_backward(const float* x, const float* dy, const float* relu_y, const float* W, int64_t M, int K, int N,
                             int precision, void* workspace, size_t workspace_bytes, float* dx, float* dW, float* db,
                             void* stream);
/* Elementwise helpers of the same path: mode 0 dx = dy * (y > 0) (ReLU backward), mode 1 dx = dy * y * (1 - y) (sigmoid
 * backward), mode 2 dx = sigmoid(dy) (forward; y unused). */
int tgtc_s2d_activation(const float* dy, const float* y, int64_t n, int mode, float* dx, void* stream);

/* trans_test.py:172-173: bilinear resize, align_corners=True.  in [C,h,w] -> out [C,H,W]. */
int tgtc_s2d_resize_bilinear(const float* in, int C, int h, int w, float* out, int H, int W, void* stream);
/* trans_test.py:176: rows = hs_nchw.reshape(-1,512); feature = [rows.mean(0), rows.var(0)] (unbiased) -> [1024].
 * hs is given token-major [n_tokens,512]; the reference's (c,h,w)-ordered flattening is reproduced exactly. */
int tgtc_s2d_style_feature(const float* hs_tokens, int n_tokens, float* feature, void* stream);
/* token-major [n,C] <-> channel-major [C,n] */
int tgtc_s2d_tokens_to_nchw(const float* tokens, int n, int C, float* out, void* stream);
int tgtc_s2d_nchw_to_tokens(const float* in, int n, int C, float* tokens, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TGTC_STYLE2D_H */
