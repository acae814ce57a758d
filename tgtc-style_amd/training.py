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
                      sigma_noise_std=1.0, jitter=None, as_float=True):
    """-> dict(loss, loss_rgb, loss_rgb_fine); parameters of both networks updated in place.  as_float=False returns the
    losses as detached device scalars WITHOUT synchronising: the reference reads them only every i_print iterations
    (train_tgtcs.py:257-266), and a float() per iteration serialises the host with the GPU."""
    val = (lambda t: float(t.detach())) if as_float else (lambda t: t.detach())
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
        out['loss_rgb_fine'] = val(loss_rgb_fine)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    out.update(loss=val(loss), loss_rgb=val(loss_rgb))
    return out


def L2_norm(x):
    return torch.sqrt(torch.sum(x ** 2) + 1e-8)          # utils.py:459


def cosine_similarity(a, b):
    """VGGNet.py:204-210: per-row cosine of two [R, C] tensors with the reference's +1e-8 on the norms."""
    a_n = a / (torch.norm(a, dim=1, keepdim=True) + 1e-8)
    b_n = b / (torch.norm(b, dim=1, keepdim=True) + 1e-8)
    return torch.sum(a_n * b_n, dim=1)


class CoherenceState:
    """What the coherence term of `Style_train` carries from one iteration to the next (train_tgtcs.py:346-349): the
    stylised coarse / fine colours of the previous frame-ordered batch (x, y), the un-stylised colours of the same rays
    (x_origin) and the batch counter `cnt` that restarts the comparison every `frame_num` batches (:396-403, :451-458).

    The reference leaves x and y attached to the previous iteration's graph and calls `backward(retain_graph=True)`; the
    optimiser has by then updated, in place, weights that graph saved, so autograd raises at the second iteration (current
    PyTorch: "one of the variables needed for gradient computation has been modified by an inplace operation").  Here the
    carried tensors are values (detached): the loss is the reference's number, its gradient flows through the current
    batch."""

    def __init__(self, frame_num):
        self.frame_num, self.cnt = frame_num, 0
        self.x = self.y = self.x_origin = None

    def coarse(self, rgb2, rgb_origin2):
        loss = rgb2.new_zeros(())
        if self.cnt != self.frame_num and self.cnt != 0:
            loss = L2_norm(cosine_similarity(rgb2, self.x) - cosine_similarity(rgb_origin2, self.x_origin))
        self.x, self.x_origin = rgb2.detach(), rgb_origin2.detach()
        return loss

    def fine(self, rgb_fine2, rgb_origin2):
        """x_origin has just been replaced by THIS batch's colours (:399-403 run first), so the second cosine compares
        rgb_origin2 with itself, as in the reference (:456)."""
        loss = rgb_fine2.new_zeros(())
        if self.cnt == self.frame_num:
            self.cnt = 1
        else:
            if self.cnt != 0:
                loss = L2_norm(cosine_similarity(rgb_fine2, self.y) - cosine_similarity(rgb_origin2, self.x_origin))
            self.cnt += 1
        self.y = rgb_fine2.detach()
        return loss


def style_train_step(model, model_fine, concat_model, style_model, latents, optimizer, rays_o, rays_d, rgb_gt, style_ids,
                     frame_ids, N_samples, N_samples_fine, near, far, sigma_noise_std=1.0, rgb_loss_lambda=1.0,
                     logp_loss_lambda=0.0, data_type='llff', jitter=None, coherence=None, coh_batch=None,
                     loss_coh_lambda=0.0, as_float=True):
    """One `Style_train` iteration (train_tgtcs.py:352-482): the shuffled batch's rendering, pixel and -log p terms and,
    with `coherence` (a CoherenceState) and `coh_batch` (dict rays_o, rays_d, rgb_origin, style_id, frame_id [, jitter]: the
    frame-ordered batch of `loss_coh_get_batch`, :366-370), the cosine-coherence term between consecutive frame-ordered
    batches, weighted by `loss_coh_lambda` (args.loss_coh_lambda; the reference drops the term after step 122 000, :472-479).
    The NeRF networks are frozen feature extractors (their fused forward, no graph); the concat / style MLPs and the latent
    table -- marked `.trainable()` -- receive gradients through the HIP dense layers, the latent gather and compositing.
    The VGG content / style losses of the reference's later stages (:484-560) are not part of it."""
    def styled(ro, rd, sid, fid, jit):
        R = ro.shape[0]
        z = latents(style_ids=sid, frame_ids=fid, type=data_type)
        zbar = torch.mean(z, dim=1, keepdim=True)
        L = z.shape[-1]

        def one_pass(nerf, pts, n):
            with torch.no_grad():
                ret = nerf(pts=pts, dirs=rd.unsqueeze(1).expand([R, n, 3]))
            cf = concat_model(x=ret['pts'], latent=z.unsqueeze(1).expand([R, n, L]))['concat_features']
            both = torch.cat((ret['base_remap'], cf), dim=-1)
            rgb = style_model(x=ret['pts'], concated=both, latent=zbar.unsqueeze(2).expand([R, n, L]))['rgb']
            return rgb, ret['sigma']

        pts, ts = utils.sampling_pts_uniform(rays_o=ro, rays_d=rd, N_samples=N_samples, near=near, far=far, perturb=True, jitter=jit)
        rgb, sigma = one_pass(model, pts, N_samples)
        rgb_exp, _, weights = utils.alpha_composition(rgb, sigma, ts, sigma_noise_std)
        rgb_exp_fine = None
        if N_samples_fine > 0:
            pts_f, ts_f = utils.sampling_pts_fine_torch(ro, rd, ts, weights.detach(), N_samples_fine)
            rgb, sigma = one_pass(model_fine, pts_f, N_samples + N_samples_fine)
            rgb_exp_fine, _, _ = utils.alpha_composition(rgb, sigma, ts_f, sigma_noise_std)
        return rgb_exp, rgb_exp_fine

    loss_coh = None
    if coherence is not None and coh_batch is not None:          # :366-403, :436-458
        rgb2, rgb_fine2 = styled(coh_batch['rays_o'], coh_batch['rays_d'], coh_batch['style_id'], coh_batch['frame_id'],
                                 coh_batch.get('jitter'))
        loss_coh = coherence.coarse(rgb2, coh_batch['rgb_origin'])
        if rgb_fine2 is not None:
            loss_coh = loss_coh + coherence.fine(rgb_fine2, coh_batch['rgb_origin'])
    rgb_exp, rgb_exp_fine = styled(rays_o, rays_d, style_ids, frame_ids, jitter)
    loss_rgb = rgb_loss_lambda * img2mse(rgb_exp, rgb_gt)
    loss_logp = logp_loss_lambda * latents.minus_logp(style_ids=style_ids, frame_ids=frame_ids, data_type=data_type)
    if rgb_exp_fine is not None:
        loss_rgb = loss_rgb + rgb_loss_lambda * img2mse(rgb_exp_fine, rgb_gt)
    loss = loss_rgb + loss_logp
    total = loss if loss_coh is None else loss + loss_coh_lambda * loss_coh     # loss_for_style (:470)
    optimizer.zero_grad()
    total.backward()
    optimizer.step()
    val = (lambda t: float(t.detach())) if as_float else (lambda t: t.detach())     # see origin_train_step
    out = {'loss': val(loss), 'loss_rgb': val(loss_rgb), 'loss_logp': val(loss_logp)}
    if loss_coh is not None:
        out['loss_coh'] = val(loss_coh)
    return out
