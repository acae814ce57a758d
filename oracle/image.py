"""Oracle: the per-frame image epilogue of the render drivers (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference rendering.py:66-71 (cal_geometry), :202-206 (render_style: eps = 1e-7, one depth plane) and
:358-361 (render_train_style: no eps, depth broadcast to three channels), and utils.py:463 (to8b = uint8 cast).
"""
import numpy as np


def frames_to_uint8(rgb, t, frames, eps=1e-7):
    """rgb float32 [frames*P,3], t float32 [frames*P] -> (uint8 [frames,P,3], uint8 [frames,P]).

    sv_t = (sv_t - min) / (max - min + eps) per frame (:69 / :205; eps is a Python float, so the arithmetic stays
    float32); np.array(x * 255, np.int32) (:71 / :207); to8b = np.array(x, dtype=np.uint8) -- a wrapping cast."""
    rgb = np.asarray(rgb, np.float32).reshape(frames, -1, 3)
    sv_t = np.asarray(t, np.float32).reshape(frames, -1)
    lo, hi = np.min(sv_t, axis=1, keepdims=True), np.max(sv_t, axis=1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        sv_t = (sv_t - lo) / ((hi - lo + eps) if eps else (hi - lo))
        sv_rgb, sv_t = np.array(rgb * 255, np.int32), np.array(sv_t * 255, np.int32)
    return sv_rgb.astype(np.uint8), sv_t.astype(np.uint8)
