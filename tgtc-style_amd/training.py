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


def style_train_step(model, model_fine, concat_model, style_model, latents, optimizer, rays_o, rays_d, rgb_gt, style_ids,
                     frame_ids, N_samples, N_samples_fine, near, far, sigma_noise_std=1.0, rgb_loss_lambda=1.0,
                     logp_loss_lambda=0.0, data_type='llff', jitter=None):
    """The rendering, pixel and -log p terms of one `Style_train` iteration (train_tgtcs.py:404-482): the NeRF networks are
    frozen feature extractors here (their fused forward, no graph), the concat / style MLPs and the latent table -- marked
    `.trainable()` -- receive gradients through the HIP dense layers, the latent gather and compositing.  The VGG content /
    style losses and the coherence term of the reference's later stages (:394-401, :456, :484-560) are not part of it."""
    R = rays_o.shape[0]
    z = latents(style_ids=style_ids, frame_ids=frame_ids, type=data_type)
    zbar = torch.mean(z, dim=1, keepdim=True)
    L = z.shape[-1]

    def one_pass(nerf, pts, n):
        with torch.no_grad():
            ret = nerf(pts=pts, dirs=rays_d.unsqueeze(1).expand([R, n, 3]))
        cf = concat_model(x=ret['pts'], latent=z.unsqueeze(1).expand([R, n, L]))['concat_features']
        both = torch.cat((ret['base_remap'], cf), dim=-1)
        rgb = style_model(x=ret['pts'], concated=both, latent=zbar.unsqueeze(2).expand([R, n, L]))['rgb']
        return rgb, ret['sigma']

    pts, ts = utils.sampling_pts_uniform(rays_o=rays_o, rays_d=rays_d, N_samples=N_samples, near=near, far=far, perturb=True,
                                         jitter=jitter)
    rgb, sigma = one_pass(model, pts, N_samples)
    rgb_exp, _, weights = utils.alpha_composition(rgb, sigma, ts, sigma_noise_std)
    loss_rgb = rgb_loss_lambda * img2mse(rgb_exp, rgb_gt)
    loss_logp = logp_loss_lambda * latents.minus_logp(style_ids=style_ids, frame_ids=frame_ids, data_type=data_type)
    if N_samples_fine > 0:
        pts_f, ts_f = utils.sampling_pts_fine_torch(rays_o, rays_d, ts, weights.detach(), N_samples_fine)
        rgb, sigma = one_pass(model_fine, pts_f, N_samples + N_samples_fine)
        rgb_exp_fine, _, _ = utils.alpha_composition(rgb, sigma, ts_f, sigma_noise_std)
        loss_rgb = loss_rgb + rgb_loss_lambda * img2mse(rgb_exp_fine, rgb_gt)
    loss = loss_rgb + loss_logp
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return {'loss': float(loss.detach()), 'loss_rgb': float(loss_rgb.detach()), 'loss_logp': float(loss_logp.detach())}
