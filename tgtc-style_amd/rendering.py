"""Render drivers.

`RayRenderer` is the MI355X-native fast path: one C-ABI call per batch of rays enqueues the whole
coarse -> fine chain (reference rendering.py:27-51 for plain, :118-178 for stylised) on the current
stream, with every intermediate kept in a preallocated device workspace.

`cal_geometry`, `render_style` and `render_train_style` keep the reference's signatures
(rendering.py:5, :93-94, :242-243) for drop-in use from `train_tgtcs.py`-style drivers.
"""
import os

import numpy as np
import torch

from . import hip


class RayRenderer:
    """Fused renderer over packed networks.

    coarse / fine: `models.StyleNerf` modules (or anything with `.packed()` returning a hip.Net).
    style: optional `models.StylePair` for the stylised chain.
    """

    def __init__(self, coarse, fine, style=None):
        self.coarse, self.fine, self.style = coarse, fine, style
        self._ws = None

    def _workspace(self, R, nc, nf, device):
        need = hip.load().tgtc_render_workspace_bytes(R, nc, nf)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def render(self, rays_o, rays_d, n_coarse, n_fine, near=0., far=1., jitter=None, z=None, want_coarse=False):
        """rays_o, rays_d float64 [R,3] on the GPU -> dict rgb [R,3], t [R] (+ rgb_coarse, t_coarse)."""
        hip.require_gpu(rays_o, rays_d)
        lib = hip.load()
        if n_fine <= 0:
            raise ValueError("N_samples_fine must be > 0 (the reference render paths dereference None otherwise)")
        rays_o = rays_o.to(torch.float64).contiguous()
        rays_d = rays_d.to(torch.float64).contiguous()
        R, dev = rays_o.shape[0], rays_o.device
        ws = self._workspace(R, n_coarse, n_fine, dev)
        rgb = torch.empty(R, 3, device=dev, dtype=torch.float32)
        t = torch.empty(R, device=dev, dtype=torch.float32)
        rgb_c = torch.empty(R, 3, device=dev, dtype=torch.float32) if want_coarse else None
        t_c = torch.empty(R, device=dev, dtype=torch.float32) if want_coarse else None
        if jitter is not None:
            jitter = jitter.to(torch.float32).contiguous()
        if self.style is None or z is None:
            hip.check(lib.tgtc_render_rays_plain(self.coarse.packed().handle, self.fine.packed().handle,
                                                 hip.ptr(rays_o), hip.ptr(rays_d), R, n_coarse, n_fine, float(near),
                                                 float(far), hip.ptr(jitter), hip.ptr(ws), ws.numel(), hip.ptr(rgb),
                                                 hip.ptr(t), hip.ptr(rgb_c), hip.ptr(t_c), hip.stream()))
        else:
            z = z.to(torch.float32).contiguous()
            hip.check(lib.tgtc_render_rays_styled(self.coarse.packed().handle, self.fine.packed().handle,
                                                  self.style.packed().handle, hip.ptr(rays_o), hip.ptr(rays_d),
                                                  hip.ptr(z), R, n_coarse, n_fine, float(near), float(far),
                                                  hip.ptr(jitter), hip.ptr(ws), ws.numel(), hip.ptr(rgb), hip.ptr(t),
                                                  hip.ptr(rgb_c), hip.ptr(t_c), hip.stream()))
        out = {"rgb": rgb, "t": t}
        if want_coarse:
            out["rgb_coarse"], out["t_coarse"] = rgb_c, t_c
        return out
