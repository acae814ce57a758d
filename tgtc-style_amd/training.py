"""One iteration of the reference's `Origin_train` body (train_tgtcs.py:226-254) on the HIP operators: stratified coarse
sampling with jitter, coarse NeRF, compositing with the density-noise regulariser, inverse-CDF fine sampling on the
(detached) coarse weights, fine NeRF, compositing, the two MSE losses, backward, optimiser step.

The networks must be marked `.trainable()` (models.StyleNerf): their forward then runs layer by layer on the differentiable
HIP dense layers (autograd_ops.py) and `utils.alpha_composition` carries its own backward kernel.  The samplers carry no
gradient, as in the reference (utils.py:562-579 detach).  The loops around this body -- data loading, learning-rate decay,
checkpoint cadence (checkpoints.py writes the files), the style stages -- are the reference's own training driver and stay
outside this build."""
import torch

from . import utils


def img2mse(x, y):
    return torch.mean((x - y) ** 2)         # utils.py img2mse


def origin_train_step(model, model_fine, optimizer, rays_o, rays_d, rgb_gt, N_samples, N_samples_fine, near, far,
                      sigma_noise_std=1.0, jitter=None):
    """-> dict(loss, loss_rgb, loss_rgb_fine) as Python floats; parameters of both networks updated in place."""
    R = rays_o.shape[0]
    pts, ts = utils.sampling_pts_uniform(rays_o=rays_o, rays_d=rays_d, N_samples=N_samples, near=near, far=far, perturb=True,
                                         jitter=jitter)
    ret = model(pts=pts, dirs=rays_d.unsqueeze(1).expand([R, N_samples, 3]))
    rgb_exp, _, weights = utils.alpha_composition(ret['rgb'], ret['sigma'], ts, sigma_noise_std)
    loss_rgb = img2mse(rgb_gt, rgb_exp)
    loss, out = loss_rgb, {}
    if N_samples_fine > 0:
        pts_fine, ts_fine = utils.sampling_pts_fine_torch(rays_o, rays_d, ts, weights.detach(), N_samples_fine)
        n = N_samples + N_samples_fine
        ret = model_fine(pts=pts_fine, dirs=rays_d.unsqueeze(1).expand([R, n, 3]))
        rgb_exp_fine, _, _ = utils.alpha_composition(ret['rgb'], ret['sigma'], ts_fine, sigma_noise_std)
        loss_rgb_fine = img2mse(rgb_gt, rgb_exp_fine)
        loss = loss + loss_rgb_fine
        out['loss_rgb_fine'] = float(loss_rgb_fine.detach())
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    out.update(loss=float(loss.detach()), loss_rgb=float(loss_rgb.detach()))
    return out
