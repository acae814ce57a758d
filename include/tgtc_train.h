/* libtgtc_hip.so -- fused training path of the NeRF MLP.
 *
 * The reference trains through PyTorch autograd: `Origin_train` (train_tgtcs.py:218-309) runs MLP_style.forward
 * (models.py:95-117 inside StyleNerf.forward :216-223) on every sample of a batch and `loss.backward()` walks the recorded
 * graph.  These entry points are what a binding would call in place of that forward / backward pair for one StyleNerf:
 * same inputs (sample points and view directions), same outputs (rgb, sigma), and on the way back the gradients of the
 * twelve nn.Linear weight / bias tensors given dL/d rgb and dL/d sigma.  Same conventions as tgtc_hip.h: int return codes,
 * tgtc_last_error(), device pointers, caller-allocated outputs and workspace, a hipStream_t, no implicit synchronisation.
 *
 * `params` / `grads`: HOST arrays of 24 DEVICE pointers in the order of MLP_style.layers (models.py:93) --
 * base_layers[0..7], sigma_layer, base_remap_layer, rgb_layers[0], rgb_layers[1] -- weight then bias for each
 * (weight [out, in] row-major fp32 as nn.Linear stores it).  The weights are read on the device at every call: an
 * optimiser may update them in place between calls.  Network shape: D = 8, W = 256, skip at 4, view directions, ReLU. */
#ifndef TGTC_TRAIN_H
#define TGTC_TRAIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgtc_trainer tgtc_trainer; /* opaque: pack maps and the packed streams of ONE network */

int tgtc_trainer_create(tgtc_trainer** out);
int tgtc_trainer_destroy(tgtc_trainer* trainer);
/* bytes of workspace for M samples: activations of every layer (fp16 hi/lo), their ReLU gates, pre-activation gradients */
size_t tgtc_trainer_workspace_bytes(int64_t M);
/* forward of StyleNerf on pts / dirs double [M,3]: rgb float [M,3], sigma float [M]; leaves in `workspace` what backward needs */
int tgtc_trainer_forward(tgtc_trainer* trainer, const float* const* params, const double* pts, const double* dirs, int64_t M,
                         void* workspace, size_t workspace_bytes, float* rgb, float* sigma, void* stream);
/* backward of the same call (same M, same workspace, untouched in between): rgb = the forward's output, d_rgb [M,3] and
 * d_sigma [M] = dL/d outputs; grads[i] (shapes of params[i]) are OVERWRITTEN with dL/d params[i]. */
int tgtc_trainer_backward(tgtc_trainer* trainer, const float* const* params, const float* rgb, const float* d_rgb,
                          const float* d_sigma, int64_t M, void* workspace, size_t workspace_bytes, float* const* grads,
                          void* stream);
/* synchronises `stream`; non-zero (with tgtc_last_error) if the last backward overflowed its fp16 operand range */
int tgtc_trainer_status(tgtc_trainer* trainer, void* stream);
/* Overflow guard: a backward whose scaled gradients left the fp16 range (growth above ~2^7 across one transposed layer), or
 * that produced any non-finite gradient for whatever reason (a forward that overflowed fp16, non-finite dL/d outputs),
 * ZERO-FILLS grads[0..23] on the device before it returns control to the stream -- an optimiser step on them cannot write
 * inf / NaN into the weights -- and counts the event.  *count = such backwards since trainer_create (synchronises `stream`;
 * read it at the reference's i_print cadence, train_tgtcs.py:257-266, not per iteration). */
int tgtc_trainer_overflows(tgtc_trainer* trainer, void* stream, unsigned* count);

/* Workspace: about 20 KB per sample -- two fp16 planes of 2 528 activations, 2 448 fp32 pre-activation gradients, one
 * 64-bit gate word per gated layer, 16-sample tile and lane -- e.g. 2.6 GB at M = 131 072 (1 024 rays x 128 samples of the
 * fine network).  It belongs to ONE forward / backward pair: a second forward before the backward of the first needs its
 * own workspace (the reference's batchify, utils.py:435-456, makes two forwards before one backward). */

#ifdef __cplusplus
}
#endif
#endif
