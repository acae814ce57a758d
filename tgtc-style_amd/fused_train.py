"""Fused training path of the NeRF MLP (include/tgtc_train.h, csrc/mlp_train.hip): StyleNerf.forward and its backward as
three HIP launches per network instead of one GEMM launch per product (autograd_ops.py).

    trainer = NerfTrainer()                                  # one per network (coarse, fine)
    rgb, sigma = trainer.apply(module.net, pts, dirs)        # differentiable w.r.t. the 24 parameters of MLP_style

`apply` is a torch.autograd.Function: forward = tgtc_trainer_forward (gather-pack of the current weights + the fused PE /
12-layer kernel with activation stash), backward = tgtc_trainer_backward (input-gradient chain + weight gradients).  The
sample points carry no gradient, as in the reference's training bodies (the samplers detach, utils.py:562-579)."""
import ctypes

import torch

from . import hip

c_int, c_int64, c_size_t, c_void_p = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p
_PTRS = ctypes.POINTER(c_void_p)

SIGNATURES = {
    "tgtc_trainer_create": [ctypes.POINTER(c_void_p)],
    "tgtc_trainer_destroy": [c_void_p],
    "tgtc_trainer_workspace_bytes": [c_int64],
    "tgtc_trainer_forward": [c_void_p, _PTRS, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p],
    "tgtc_trainer_backward": [c_void_p, _PTRS, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, _PTRS, c_void_p],
    "tgtc_trainer_status": [c_void_p, c_void_p],
    "tgtc_trainer_overflows": [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_uint)],
}
hip.register(SIGNATURES, {"tgtc_trainer_workspace_bytes": c_size_t})


def mlp_parameters(net):
    """The 24 tensors in the order of MLP_style.layers (models.py:93): weight, bias of base_layers[0..7], sigma_layer,
    base_remap_layer, rgb_layers[0], rgb_layers[1]."""
    layers = list(net.base_layers) + [net.sigma_layer, net.base_remap_layer] + list(net.rgb_layers)
    out = []
    for layer in layers:
        out += [layer.weight, layer.bias]
    return out


def _table(tensors):
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


MAX_WORKSPACE_BYTES = 96 << 30     # a third of the card: beyond it the caller should chunk the batch (20 KB per sample)


class NerfTrainer:
    """Workspaces (activation stash, gate words, pre-activation gradients: ~20 KB per sample, include/tgtc_train.h) belong
    to ONE forward / backward pair: `lease` hands one out per forward, the autograd node keeps it until its backward has
    run and `release` returns it to the pool.  Two forwards before one backward -- the reference's batchify
    (utils.py:435-456), gradient accumulation -- therefore hold two workspaces; `drop_workspaces` frees the pool
    (StyleNerf.trainable(False))."""

    def __init__(self):
        self.lib = hip.load()
        h = c_void_p()
        hip.check(self.lib.tgtc_trainer_create(ctypes.byref(h)))
        self.handle = h
        self._pool = []

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tgtc_trainer_destroy(self.handle)
        except Exception:
            pass

    def lease(self, M, device):
        """a workspace for one forward / backward pair of M samples"""
        need = int(self.lib.tgtc_trainer_workspace_bytes(M))
        if need > MAX_WORKSPACE_BYTES:
            raise ValueError("fused training workspace for %d samples = %.1f GiB (about 20 KB per sample): chunk the batch "
                             "(utils.batchify) or train on the per-layer path (.trainable(fused=False))" % (M, need / 2.0 ** 30))
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        for i, ws in enumerate(self._pool):
            if ws.numel() >= need and ws.device == device:
                return self._pool.pop(i)
        return torch.empty(need, dtype=torch.uint8, device=device)

    def release(self, ws):
        if ws is not None and len(self._pool) < 4:
            self._pool.append(ws)

    def drop_workspaces(self):
        self._pool = []

    def forward(self, params, pts, dirs, ws):
        M, dev = pts.shape[0], pts.device
        rgb = torch.empty(M, 3, device=dev, dtype=torch.float32)
        sigma = torch.empty(M, device=dev, dtype=torch.float32)
        hip.check(self.lib.tgtc_trainer_forward(self.handle, _table(params), hip.ptr(pts), hip.ptr(dirs), M, hip.ptr(ws), ws.numel(),
                                                hip.ptr(rgb), hip.ptr(sigma), hip.stream()))
        return rgb, sigma

    def backward(self, params, rgb, d_rgb, d_sigma, ws):
        """gradients of the 24 parameters; zero-filled by the library's overflow guard when the scaled input gradients left
        the fp16 range (include/tgtc_train.h, `overflows`)"""
        M = rgb.shape[0]
        flat = torch.empty(sum(p.numel() for p in params), device=rgb.device, dtype=torch.float32)     # back to back: one zero-fill
        grads, at = [], 0
        for p in params:
            grads.append(flat[at:at + p.numel()].view(p.shape))
            at += p.numel()
        hip.check(self.lib.tgtc_trainer_backward(self.handle, _table(params), hip.ptr(rgb), hip.ptr(d_rgb), hip.ptr(d_sigma), M,
                                                 hip.ptr(ws), ws.numel(), _table(grads), hip.stream()))
        return grads

    def status(self):
        hip.check(self.lib.tgtc_trainer_status(self.handle, hip.stream()))

    def overflows(self):
        """backwards since creation whose gradients the overflow guard zero-filled (synchronises; read it at the reference's
        i_print cadence, train_tgtcs.py:257-266)"""
        n = ctypes.c_uint(0)
        hip.check(self.lib.tgtc_trainer_overflows(self.handle, hip.stream(), ctypes.byref(n)))
        return int(n.value)

    def apply(self, net, pts, dirs):
        """pts, dirs: [..., 3] (any float dtype) -> rgb [..., 3], sigma [...]"""
        lead = pts.shape[:-1]
        p = pts.reshape(-1, 3).to(torch.float64).contiguous()
        d = dirs.expand(*lead, 3).reshape(-1, 3).to(torch.float64).contiguous()
        rgb, sigma = _NerfTrainFn.apply(self, p, d, *mlp_parameters(net))
        return rgb.reshape(*lead, 3), sigma.reshape(*lead)


class _NerfTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trainer, pts, dirs, *params):
        hip.require_gpu(pts, dirs, *params)
        ps = [p.detach().float().contiguous() for p in params]
        ws = trainer.lease(pts.shape[0], pts.device)
        rgb, sigma = trainer.forward(ps, pts, dirs, ws)
        ctx.trainer, ctx.ps, ctx.ws = trainer, ps, ws          # the stash of THIS call, until its backward has run
        ctx.save_for_backward(rgb)
        ctx.mark_non_differentiable()
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        (rgb,) = ctx.saved_tensors
        M = rgb.shape[0]
        d_rgb = torch.zeros_like(rgb) if d_rgb is None else d_rgb.float().contiguous()
        d_sigma = torch.zeros(M, device=rgb.device) if d_sigma is None else d_sigma.float().contiguous()
        if ctx.ws is None:
            raise RuntimeError("fused NeRF training: backward called twice on one forward (retain_graph): the activation stash "
                               "was returned to the pool after the first backward")
        grads = ctx.trainer.backward(ctx.ps, rgb, d_rgb, d_sigma, ctx.ws)
        ctx.trainer.release(ctx.ws)
        ctx.ws = None
        return (None, None, None) + tuple(grads)
