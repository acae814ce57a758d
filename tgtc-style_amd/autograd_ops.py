"""Differentiable dense layers on the HIP GEMM kernel, for the training side of the reference (SURVEY section 8f rank 4:
`Origin_train` / `Style_train`, train_tgtcs.py:218-571, backpropagate through the NeRF MLPs).

The render path never needs these: it runs the fused kernels.  When a module's parameters require gradients its
forward switches to this layer-by-layer form -- every product still a HIP kernel (`tgtc_s2d_linear`,
`tgtc_s2d_linear_backward`: dx = dy.W, dW = dy^T.x with the sample dimension split over GEMM batches, db = column sums;
`tgtc_s2d_activation` for ReLU / sigmoid), torch only records the graph and concatenates.  It is the plain form of the
training arithmetic, not a fused one: a 196 k-sample batch costs about as much as the same layers through rocBLAS."""
import torch

from . import hip
from . import style2d  # noqa: F401  (registers the ctypes signatures of the tgtc_s2d_* entries)


def _ws(nbytes, device):
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def _split(weight, w):
    """fp16 hi / lo halves of a weight, split once per optimiser step.  The cache entry is an attribute of the
    Parameter itself -- it lives and dies with it (no id() reuse after a model is freed, nothing accumulates) -- and is
    valid for one (storage, version) of the parameter."""
    hit = getattr(weight, "_tgtc_halves", None)
    if hit is None or hit[0] != weight.data_ptr() or hit[1] != weight._version:
        lib = hip.load()
        hi = torch.empty(w.numel(), dtype=torch.float16, device=w.device)
        lo = torch.empty_like(hi)
        hip.check(lib.tgtc_s2d_split(hip.ptr(w), w.numel(), hip.ptr(hi), hip.ptr(lo), hip.stream()))
        hit = (weight.data_ptr(), weight._version, hi, lo)
        weight._tgtc_halves = hit
    return hit[2], hit[3]


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu, precision):
        lib = hip.load()
        x, w = x.float().contiguous(), weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=x.device)
        hi, lo = _split(weight, w) if K % 4 == 0 else (None, None)
        hip.check(lib.tgtc_s2d_linear_pre(hip.ptr(x), M, K, hip.ptr(w), hip.ptr(hi), hip.ptr(lo), hip.ptr(b), N, int(relu),
                                          hip.PRECISIONS[precision], hip.ptr(y), hip.stream()))
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu, ctx.precision, ctx.has_bias = relu, precision, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        x, w, y = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dy = dy.float().contiguous()        # gated by the forward's ReLU inside the backward kernels (relu_y)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(w) if need_w else None
        db = torch.empty(N, device=x.device) if need_b else None
        ws = _ws(lib.tgtc_s2d_linear_backward_workspace_bytes(M, K, N), x.device)
        hip.check(lib.tgtc_s2d_linear_backward(hip.ptr(x), hip.ptr(dy), hip.ptr(y) if ctx.relu else None, hip.ptr(w), M, K, N, hip.PRECISIONS[ctx.precision], hip.ptr(ws),
                                               ws.numel() * 4, hip.ptr(dx), hip.ptr(dw), hip.ptr(db), hip.stream()))
        return dx, dw, db, None, None


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = hip.load()
        x = x.float().contiguous()
        y = torch.empty_like(x)
        hip.check(lib.tgtc_s2d_activation(hip.ptr(x), None, x.numel(), 2, hip.ptr(y), hip.stream()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        (y,) = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx = torch.empty_like(dy)
        hip.check(lib.tgtc_s2d_activation(hip.ptr(dy), hip.ptr(y), dy.numel(), 1, hip.ptr(dx), hip.stream()))
        return dx


def linear(x, layer, relu=False, precision="fp16x3"):
    """nn.Linear `layer` (+ ReLU) on x [M,K], differentiable w.r.t. x, weight and bias."""
    return _Linear.apply(x, layer.weight, layer.bias, relu, precision)


def sigmoid(x):
    return _Sigmoid.apply(x)
