"""Import shim used ONLY when generating golden fixtures in the build container.

The reference tree (/root/reference) is pure Python/PyTorch but several of its
modules import third-party packages that are not installed here (cv2, imageio,
plyfile, pyrender, skimage, natsort, torchvision, tensorboardX, configargparse,
pytorch3d).  None of the hot-path functions touches a symbol of those packages,
so we register inert placeholder modules before importing the reference.

Nothing in this file copies reference source; it only makes `import utils`,
`import models`, ... succeed so that `gen_golden.py` can call the reference's
own functions and record their inputs/outputs as data fixtures.

This file is never imported by the product, the GPU tests, smoke() or bench.py:
/root/reference does not exist on the GPU box.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TGTC_REFERENCE_ROOT", "/root/reference")


class _Inert(types.ModuleType):
    """A module whose every attribute is another inert object."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        sub = _Inert(self.__name__ + "." + item)
        setattr(self, item, sub)
        return sub

    def __call__(self, *a, **k):  # allows decorator-style / constructor use at import time
        return self


_STUBS = [
    "cv2", "imageio", "pyrender", "skimage", "skimage.feature", "skimage.metrics",
    "tensorboardX", "configargparse", "pytorch3d", "pytorch3d.structures",
    "pytorch3d.renderer", "open3d", "colormath", "plyfile", "natsort",
    "torchvision", "torchvision.transforms", "torchvision.models", "torchvision.utils",
]


def install():
    """Register placeholders and put the reference on sys.path. Idempotent."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    saved_cvd = os.environ.get("CUDA_VISIBLE_DEVICES")
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = _Inert(name)
    sys.modules["natsort"].natsorted = sorted
    sys.modules["torchvision"]._is_tracing = lambda: False
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return saved_cvd


def restore_env(saved_cvd):
    """reference transformer.py:11 sets CUDA_VISIBLE_DEVICES as an import side effect; undo it."""
    if saved_cvd is None:
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    else:
        os.environ["CUDA_VISIBLE_DEVICES"] = saved_cvd
