"""Camera poses of an LLFF scene, host side (the ray-delivery contract up to `cps_valid`).

What the reference derives from `poses_bounds.npy` before any ray exists (load_llff.py:60-61,95-96 raw layout and
image size; :233-305 axis fix-up, bound rescale, recentring, the 120-view spiral, the hold-out view; dataset.py:101-102
the 4x4 `cps_valid`).  Pure numpy like the reference; the device side starts at `utils.gen_rays`, which takes these
poses.  Pinned to the reference's own `load_llff_data` output on a synthetic scene directory (tests/golden g11).
"""
import numpy as np


def _unit(v):
    return v / np.linalg.norm(v)


def look_frame(forward, up_hint, position):
    """[right | up | forward | position] 3x4 with orthonormalised axes (load_llff.py:121-127)."""
    f = _unit(forward)
    r = _unit(np.cross(up_hint, f))
    return np.stack([r, _unit(np.cross(f, r)), f, position], axis=1)


def average_pose(poses):
    """Mean camera of [N,3,5] poses, hwf column carried over from the first one (load_llff.py:133-143)."""
    frame = look_frame(poses[:, :3, 2].sum(0), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([frame, poses[0, :3, -1:]], axis=1)


def recentre(poses):
    """Express every pose in the frame of the average camera (load_llff.py:160-174)."""
    def homogeneous(p34):
        last = np.broadcast_to(np.array([0, 0, 0, 1.], p34.dtype), p34.shape[:-2] + (1, 4))
        return np.concatenate([p34, last], axis=-2)

    to_avg = np.linalg.inv(homogeneous(average_pose(poses)[:3, :4]))
    out = poses.copy()
    out[:, :3, :4] = (to_avg @ homogeneous(poses[:, :3, :4]))[:, :3, :4]
    return out


def spiral_path(centre_pose, up, radii, focus, z_rate, rotations, n_views):
    """`n_views` cameras on a spiral around `centre_pose`, all looking at the point `focus` in front of it
    (load_llff.py:145-154)."""
    frame, hwf = centre_pose[:3, :4], centre_pose[:, 4:5]
    radii = np.append(np.asarray(radii, dtype=np.float64), 1.0)
    target = frame @ np.array([0, 0, -focus, 1.])
    views = []
    for theta in np.linspace(0., 2. * np.pi * rotations, n_views + 1)[:-1]:
        eye = frame @ (np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * z_rate), 1.]) * radii)
        views.append(np.concatenate([look_frame(eye - target, up, eye), hwf], axis=1))
    return views


def raw_poses(poses_arr, image_hw, factor):
    """poses_bounds.npy [N,17] -> (poses [3,5,N], bounds [2,N]) with the hwf column set to the loaded image size and
    the focal length scaled by 1/factor (load_llff.py:60-61, 95-96)."""
    poses = np.ascontiguousarray(poses_arr[:, :15].reshape(-1, 3, 5).transpose(1, 2, 0))
    poses[0, 4, :], poses[1, 4, :] = image_hw
    poses[2, 4, :] *= 1. / factor
    return poses, np.ascontiguousarray(poses_arr[:, 15:].T)


def scene_poses(poses_arr, image_hw, factor=8, recenter=True, bd_factor=.75, n_views=120, n_rots=2):
    """load_llff_data without the images (spherify=False, path_zflat=False -- the flags dataset.py:69 uses; the
    reference's own path_zflat branch raises on current numpy, load_llff.py:283).

    Returns dict(poses [N,3,5] f32, bds [N,2] f32, render_poses [n_views,3,5] f32, i_test, hwf)."""
    poses, bds = raw_poses(np.asarray(poses_arr, np.float64), image_hw, factor)
    # LLFF stores [down, right, back]; the renderer wants [right, up, back] (load_llff.py:239)
    poses = np.concatenate([poses[:, 1:2], -poses[:, 0:1], poses[:, 2:]], axis=1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    scale = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= scale
    bds *= scale
    if recenter:
        poses = recentre(poses)
    centre = average_pose(poses)
    up = _unit(poses[:, :3, 1].sum(0))
    near, far = bds.min() * .9, bds.max() * 5.
    focus = 1. / (.25 / near + .75 / far)                      # dt = .75 between the near and far bounds in disparity
    radii = np.percentile(np.abs(poses[:, :3, 3]), 90, axis=0)
    render = np.array(spiral_path(centre, up, radii, focus, .5, n_rots, n_views)).astype(np.float32)
    centre = average_pose(poses)
    i_test = int(np.argmin(np.square(centre[:3, 3] - poses[:, :3, 3]).sum(-1)))
    return {"poses": poses.astype(np.float32), "bds": bds, "render_poses": render, "i_test": i_test,
            "hwf": poses[0, :3, -1]}


def valid_camera_poses(render_poses):
    """[N,3,5] spiral poses -> the [N,4,4] `cps_valid` the datasets hand to the ray generator (dataset.py:101-102)."""
    cps = np.zeros((render_poses.shape[0], 4, 4), render_poses.dtype)
    cps[:, :3, :4] = render_poses[:, :3, :4]
    cps[:, 3, 3] = 1.
    return cps
