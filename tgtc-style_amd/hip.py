"""ctypes binding of libtgtc_hip.so (include/tgtc_hip.h).

PyTorch is used for device memory and streams only: every call passes raw `data_ptr()`s and the
current HIP stream.  There is NO CPU fallback: if the library is missing or a call fails, a
RuntimeError is raised.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TGTC_LIB") or os.path.join(_HERE, "csrc", "libtgtc_hip.so")   # TGTC_LIB: development builds

PREC_FP16X3 = 0   # split-fp16, fp32-equivalent (parity mode)
PREC_FP16 = 1     # single fp16 MFMA product (fast mode)
PREC_FP16_FP6 = 2  # fp16 product + two block-scaled fp6 correction products (NeRF nets only)
PRECISIONS = {"fp16x3": PREC_FP16X3, "fp16": PREC_FP16, "fp16mx": PREC_FP16_FP6}

_lib = None

c_void_p, c_int, c_int64, c_float, c_double, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                         ctypes.c_float, ctypes.c_double, ctypes.c_size_t)


class Linear(ctypes.Structure):
    """tgtc_linear: host pointers to an nn.Linear's weight [out,in] and bias [out]."""
    _fields_ = [("weight", c_void_p), ("bias", c_void_p), ("out_features", ctypes.c_int32),
                ("in_features", ctypes.c_int32)]


# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "tgtc_version": [],
    "tgtc_last_error": [],
    "tgtc_gen_rays": [c_int, c_int, c_double, c_double, c_double, c_double, c_void_p, c_int, c_int, c_double,
                      c_int64, c_int64, c_void_p, c_void_p, c_void_p],
    "tgtc_sample_coarse": [c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_posenc": [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p],
    "tgtc_nerf_create": [ctypes.POINTER(Linear), c_int, c_int, ctypes.POINTER(c_void_p)],
    "tgtc_net_destroy": [c_void_p],
    "tgtc_net_precision": [c_void_p],
    "tgtc_nerf_forward": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_nerf_mlp_forward": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_nerf_forward_rays": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    "tgtc_composite": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_composite_train": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_composite_backward": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p],
    "tgtc_sample_fine": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "tgtc_render_workspace_bytes": [c_int64, c_int, c_int],
    "tgtc_render_rays_plain": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float,
                               c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_render_rays_plain_chain": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float,
                                     c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_render_rays_plain_fused": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p],
    "tgtc_image_epilogue": [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p, c_void_p],
    "tgtc_latents_forward": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_float, c_int,
                             c_void_p, c_void_p],
    "tgtc_latents_backward": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_float, c_int, c_void_p, c_void_p, c_void_p],
}
_RESTYPES = {"tgtc_last_error": ctypes.c_char_p, "tgtc_render_workspace_bytes": c_size_t}
_OPTIONAL = {
    "tgtc_style_create": [ctypes.POINTER(Linear), c_int, ctypes.POINTER(Linear), c_int, c_int, ctypes.POINTER(c_void_p)],
    "tgtc_concat_mlp_forward": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    "tgtc_style_mlp_forward": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    "tgtc_styled_forward_rays": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                 c_void_p, c_void_p, c_void_p],
    "tgtc_render_rays_styled": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p],
    "tgtc_render_rays_styled_chain": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                      c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p],
}


def register(signatures, restypes=None):
    """Add the entry points of another header (include/tgtc_style2d.h) to the binding table."""
    _SIGNATURES.update(signatures)
    _RESTYPES.update(restypes or {})
    if _lib is not None:
        for name, argtypes in signatures.items():
            fn = getattr(_lib, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)


def header_symbols():
    """Every entry point include/tgtc_hip.h declares (used by the CPU export test)."""
    return sorted(list(_SIGNATURES) + list(_OPTIONAL))


def load():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libtgtc_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C tgtc-style_amd/csrc` (expected at %s)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for table in (_SIGNATURES, _OPTIONAL):
        for name, argtypes in table.items():
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
    _lib = lib
    return lib


def missing_symbols():
    """Header symbols the built library does not export (must be empty)."""
    lib = ctypes.CDLL(LIB_PATH)
    return [n for n in header_symbols() if not hasattr(lib, n)]


def check(rc):
    if rc != 0:
        raise RuntimeError("libtgtc_hip: error %d: %s" % (rc, load().tgtc_last_error().decode()))


def ptr(t):
    """Device (or host) pointer of a tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "tgtc ops need contiguous tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    if not torch.cuda.is_available():
        raise RuntimeError("tgtc_style_amd: no GPU visible; the HIP path has no CPU fallback")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("tgtc_style_amd: expected a CUDA/HIP tensor, got device %s" % t.device)


def make_linears(pairs):
    """[(weight, bias), ...] (tensors or arrays) -> (ctypes array of tgtc_linear, keepalive list)."""
    arr = (Linear * len(pairs))()
    keep = []
    for i, (w, b) in enumerate(pairs):
        w = np.ascontiguousarray(w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else w, dtype=np.float32)
        b = np.ascontiguousarray(b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b, dtype=np.float32)
        keep += [w, b]
        arr[i].weight = w.ctypes.data
        arr[i].bias = b.ctypes.data
        arr[i].out_features, arr[i].in_features = w.shape
    return arr, keep


class Net:
    """Owns a tgtc_net handle (packed weights resident in HBM)."""

    def __init__(self, handle, precision):
        self.handle = handle
        self.precision = precision

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.tgtc_net_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


NERF_LAYER_KEYS = (["base_layers.%d" % i for i in range(8)] +
                   ["sigma_layer", "base_remap_layer", "rgb_layers.0", "rgb_layers.1"])


def nerf_create(state, precision="fp16x3", prefix="net."):
    """Pack a StyleNerf / MLP_style state dict (reference key names) into a device-resident net."""
    require_gpu()
    lib = load()
    pairs = [(state[prefix + k + ".weight"], state[prefix + k + ".bias"]) for k in NERF_LAYER_KEYS]
    arr, keep = make_linears(pairs)
    h = c_void_p()
    check(lib.tgtc_nerf_create(arr, len(pairs), PRECISIONS[precision], ctypes.byref(h)))
    del keep
    return Net(h, precision)


def style_create(concat_state=None, style_state=None, precision="fp16x3"):
    """Pack StyleMLP_before_concat (5 linears `layers.0..4`) and / or StyleMLP_Wild_multilayers (8 linears
    `layers.0..7`) state dicts into one device-resident handle.  A missing net is packed as zeros."""
    require_gpu()
    lib = load()
    keep = []
    if concat_state is not None:
        ca, k = make_linears([(concat_state["layers.%d.weight" % i], concat_state["layers.%d.bias" % i]) for i in range(5)])
        keep.append(k)
        nc = 5
    else:
        ca, nc = None, 0
    if style_state is not None:
        sa, k = make_linears([(style_state["layers.%d.weight" % i], style_state["layers.%d.bias" % i]) for i in range(8)])
        keep.append(k)
        ns = 8
    else:
        sa, ns = None, 0
    h = c_void_p()
    check(lib.tgtc_style_create(ca, nc, sa, ns, PRECISIONS[precision], ctypes.byref(h)))
    del keep
    return Net(h, precision)
