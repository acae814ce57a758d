"""Oracle: pinhole ray generation + LLFF NDC warp (numpy, float64 like the reference).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference dataset.py:33-42 (get_rays_np) and dataset.py:44-61 (ndc_rays_np).
"""
import numpy as np


def pinhole_rays(H, W, K, c2w, pixel_alignment=True):
    """Rays through every pixel of an H x W image.

    dataset.py:33-42.  Pixel (col i, row j) -> camera-frame direction
    ((i-cx)/fx, -(j-cy)/fy, -1) (float32 grid, optional +0.5 centre, :34-37),
    rotated by c2w[:3,:3] via a broadcast multiply + sum over the last axis (:39),
    origin = c2w[:3,3] broadcast (:41).  Directions are not normalised.
    Returns (rays_o, rays_d), each [H, W, 3].
    """
    K = np.asarray(K)
    c2w = np.asarray(c2w)
    col = np.arange(W, dtype=np.float32)[None, :].repeat(H, 0)
    row = np.arange(H, dtype=np.float32)[:, None].repeat(W, 1)
    if pixel_alignment:
        col = col + .5
        row = row + .5
    cam = np.stack([(col - K[0][2]) / K[0][0],
                    -(row - K[1][2]) / K[1][1],
                    -np.ones_like(col)], axis=-1)
    rays_d = (cam[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape)
    return rays_o, rays_d


def ndc_warp(H, W, focal, near, rays_o, rays_d):
    """LLFF normalised-device-coordinate warp.  dataset.py:44-61.

    Origins are first moved to the near plane (t = -(near+oz)/dz, :46-47), then the
    six projective formulas (:50-56) are applied.
    """
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx = -1. / (W / (2. * focal))
    sy = -1. / (H / (2. * focal))
    o0 = sx * o[..., 0] / o[..., 2]
    o1 = sy * o[..., 1] / o[..., 2]
    o2 = 1. + 2. * near / o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2. * near / o[..., 2]
    return np.stack([o0, o1, o2], -1), np.stack([d0, d1, d2], -1)


def frame_rays_ndc(H, W, focal, c2w, near=1.0, pixel_alignment=False):
    """The composition the datasets use for llff scenes (dataset.py:412-433):
    K = [[f,0,W/2],[0,f,H/2],[0,0,1]], get_rays_np then ndc_rays_np(near=1), stored float64.
    Returns rays_o, rays_d as float64 [H*W, 3] in row-major pixel order.
    """
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    o, d = pinhole_rays(H, W, K, np.asarray(c2w), pixel_alignment)
    o, d = ndc_warp(H, W, focal, near, o.astype(np.float64), d.astype(np.float64))
    return o.reshape(-1, 3).astype(np.float64), d.reshape(-1, 3).astype(np.float64)
