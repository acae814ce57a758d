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

    def __init__(self, coarse, fine, style=None, fused=True):
        """fused=True: the library's fastest path -- a single persistent ray kernel where one is built, except for coarse
        fp16x3 + fine fp16mx, whose fine pass runs faster on the two-tile per-sample kernel (include/tgtc_hip.h,
        tgtc_render_rays_plain: the split path, taken when the workspace is handed over).  fused="single" asks for the single
        kernel and nothing else (tgtc_render_rays_plain_fused; plain renders only).  fused=False forces the chain of
        per-sample kernels (tgtc_render_rays_plain_chain / tgtc_render_rays_styled_chain).  All three agree to rounding, the
        network arithmetic bit for bit (tests/test_fused_gpu.py)."""
        if fused not in (True, False, "single"):
            raise ValueError("fused must be True, False or 'single'")
        self.coarse, self.fine, self.style, self.fused = coarse, fine, style, fused
        self._ws = None

    def _split_is_faster(self):
        """Mirror of the library's rule (csrc/render.hip, tgtc_render_rays_plain)."""
        return (self.coarse.packed().precision, self.fine.packed().precision) == ("fp16x3", "fp16mx")

    def _fused_shape(self, nc, nf):
        """Mirror of the library's rule (include/tgtc_hip.h, tgtc_render_rays_plain): when it renders with the single
        fused kernel no workspace is needed."""
        pc, pf = self.coarse.packed().precision, self.fine.packed().precision
        pair = (pc, pf) in (("fp16x3", "fp16x3"), ("fp16x3", "fp16mx"), ("fp16", "fp16"))
        step = 32 if pc == "fp16" else 16
        return pair and nf >= 1 and nc >= 16 and nc % step == 0 and (nc + nf) % step == 0 and nc <= 192 and nc + nf <= 256

    def _fused_styled_shape(self, nc, nf):
        """The same for tgtc_render_rays_styled: the stylised ray kernel is built for fp16x3 in all three handles."""
        precs = {self.coarse.packed().precision, self.fine.packed().precision, self.style.packed().precision}
        return precs == {"fp16x3"} and self._fused_shape(nc, nf)

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
        plain = self.style is None or z is None
        one_kernel = self.fused and not want_coarse and (self._fused_shape(n_coarse, n_fine) if plain else
                                                         self._fused_styled_shape(n_coarse, n_fine))
        if self.fused == "single":
            if not (plain and one_kernel):
                raise ValueError("fused='single': no single-kernel build for this render (precisions, sample counts, coarse image or style)")
        elif one_kernel and plain and self._split_is_faster():
            one_kernel = False              # hand the workspace over: the library takes the split path
        ws = None if one_kernel else self._workspace(R, n_coarse, n_fine, dev)
        rgb = torch.empty(R, 3, device=dev, dtype=torch.float32)
        t = torch.empty(R, device=dev, dtype=torch.float32)
        rgb_c = torch.empty(R, 3, device=dev, dtype=torch.float32) if want_coarse else None
        t_c = torch.empty(R, device=dev, dtype=torch.float32) if want_coarse else None
        if jitter is not None:
            jitter = jitter.to(torch.float32).contiguous()
        if plain and self.fused == "single":
            hip.check(lib.tgtc_render_rays_plain_fused(self.coarse.packed().handle, self.fine.packed().handle, hip.ptr(rays_o),
                                                       hip.ptr(rays_d), R, n_coarse, n_fine, float(near), float(far), hip.ptr(jitter),
                                                       hip.ptr(rgb), hip.ptr(t), hip.stream()))
        elif plain:
            fn = lib.tgtc_render_rays_plain if self.fused else lib.tgtc_render_rays_plain_chain
            hip.check(fn(self.coarse.packed().handle, self.fine.packed().handle, hip.ptr(rays_o), hip.ptr(rays_d), R,
                         n_coarse, n_fine, float(near), float(far), hip.ptr(jitter), hip.ptr(ws),
                         0 if ws is None else ws.numel(), hip.ptr(rgb), hip.ptr(t), hip.ptr(rgb_c), hip.ptr(t_c),
                         hip.stream()))
        else:
            z = z.to(torch.float32).contiguous()
            fn = lib.tgtc_render_rays_styled if self.fused else lib.tgtc_render_rays_styled_chain
            hip.check(fn(self.coarse.packed().handle, self.fine.packed().handle, self.style.packed().handle, hip.ptr(rays_o),
                         hip.ptr(rays_d), hip.ptr(z), R, n_coarse, n_fine, float(near), float(far), hip.ptr(jitter), hip.ptr(ws),
                         0 if ws is None else ws.numel(), hip.ptr(rgb), hip.ptr(t), hip.ptr(rgb_c), hip.ptr(t_c), hip.stream()))
        out = {"rgb": rgb, "t": t}
        if want_coarse:
            out["rgb_coarse"], out["t_coarse"] = rgb_c, t_c
        return out


# =====================================================================================================
# Drop-in drivers with the reference's signatures (rendering.py:5, :93-94, :242-243).
#
# They accept the same injected callables and the same dataloader / dataset duck types as the reference
# (a dataloader yields dicts of tensors; `dataloader.dataset` carries cps / cps_valid / hwf / near / far /
# frame_num / h / w / mode).  With HIP-backed callables from this package every stage runs on the GPU; pass
# `renderer=RayRenderer(...)` to replace the per-stage chain by the fused single-call path.
# Unlike the reference they return cleanly and require N_samples_fine > 0 (Q2).
# =====================================================================================================
def _save_png(path, arr):
    """utils.py:463 to8b = uint8 cast.  The file is encoded and written by the background writer (image_writer.py);
    the drivers drain it before they return."""
    from .image_writer import writer
    writer().save(path, arr if isinstance(arr, torch.Tensor) else np.asarray(arr).astype(np.uint8))


def _drain_images():
    from .image_writer import writer
    writer().drain()


def _to_device(batch, device):
    return {k: torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(device) for k, v in batch.items()}


def _require_fine(args):
    if not args.N_samples_fine > 0:
        raise ValueError("N_samples_fine must be > 0: the reference's render paths dereference None otherwise "
                         "(rendering.py:186; train_tgtcs.py:184)")


# ---- multi-GPU hooks of the dataset duck type (train_tgtcs.ShardedScene; absent on reference-style datasets) --------
# A rank either owns whole images (frames sharding: it renders image k iff k % world == rank and writes its files
# itself) or a contiguous pixel range of EVERY image (rays sharding: the ranks' rows are all-gathered and rank 0
# writes).  The drivers only need: the global number of the i-th image this rank completes, how many rays of an
# image it renders, and how to assemble a finished image.
def _image_id(ds, local_no):
    return ds.global_image(local_no) if hasattr(ds, 'global_image') else local_no


def _local_res(ds, res):
    return ds.rays_per_image() if hasattr(ds, 'rays_per_image') else res


def _assemble(ds, rgb, t):
    """-> (rgb [h*w,3], t [h*w], this rank writes the files)"""
    return ds.assemble(rgb, t) if hasattr(ds, 'assemble') else (rgb, t, True)


def _write_depth_rgb(sv_path, rgb, t, h, w, rgb_name, depth_name, eps=1e-7, depth_channels=1):
    """rendering.py:202-217 (:358-361 for eps = 0, three depth channels): per-image min-max depth normalisation, x255,
    int32, uint8 cast.  CUDA tensors take the device epilogue (only the 8-bit images cross PCIe); numpy arrays the
    reference's own host arithmetic."""
    if isinstance(rgb, torch.Tensor) and rgb.is_cuda:
        from . import utils
        rgb8, depth8 = utils.frames_to_uint8(rgb, t, 1, eps)       # stay on the device: the writer copies them out
        rgb8, depth8 = rgb8.reshape(h, w, 3), depth8.reshape(h, w)
        if depth_channels == 3:
            depth8 = depth8.unsqueeze(-1).expand(h, w, 3).contiguous()
    else:
        rgb, t = np.asarray(rgb, np.float32), np.asarray(t, np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            sv_t = (t - t.min()) / ((t.max() - t.min() + eps) if eps else (t.max() - t.min()))
            rgb8 = np.array(rgb.reshape(h, w, 3) * 255, np.int32).astype(np.uint8)
            depth8 = np.array(sv_t.reshape(h, w) * 255, np.int32).astype(np.uint8)
        if depth_channels == 3:
            depth8 = np.ascontiguousarray(np.broadcast_to(depth8[..., None], [h, w, 3]))
    _save_png(os.path.join(sv_path, rgb_name), rgb8)
    _save_png(os.path.join(sv_path, depth_name), depth8)


def cal_geometry(model_forward, samp_func, dataloader, args, device, sv_path=None, model_forward_fine=None,
                 samp_func_fine=None, renderer=None):
    """reference rendering.py:5-90: plain NeRF render of every ray the loader yields; writes rgb_%05d.png,
    depth_%05d.png, geometry_%05d.npz per image and geometry.npz; returns (rgb_map [F,h,w,3], t_map [F,h,w,1])."""
    from . import utils
    _require_fine(args)
    if sv_path is not None:
        os.makedirs(sv_path, exist_ok=True)
    ds = dataloader.dataset
    train = 'train' in ds.mode
    cps = ds.cps if train else ds.cps_valid
    frame_num, h, w = (ds.frame_num if train else ds.cps_valid.shape[0]), ds.h, ds.w
    res = h * w
    # frames sharding (train_tgtcs.ShardedScene): this rank renders the images k with k % world == rank, in that order
    world, rank = getattr(ds, 'world', 1), getattr(ds, 'rank', 0)
    if world > 1 and getattr(ds, 'shard', 'frames') != 'frames':
        raise ValueError("cal_geometry shards by whole frames (--shard frames): geometry_%05d.npz is a per-frame file")
    local_frames = len(range(rank, frame_num, world)) if world > 1 else frame_num
    rgb_map = np.zeros([local_frames * res, 3], np.float32)
    t_map = np.zeros([local_frames * res], np.float32)
    coor_map = np.zeros([local_frames * res, 3], np.float32)
    img_id = pixel_id = 0
    for batch in dataloader:
        b = _to_device(batch, device)
        rays_o, rays_d = b['rays_o'], b['rays_d']
        if renderer is not None:
            out = renderer.render(rays_o, rays_d, args.N_samples, args.N_samples_fine, near=ds.near, far=ds.far)
            rgb_f, t_f = out["rgb"], out["t"]
        else:
            pts, ts = samp_func(rays_o=rays_o, rays_d=rays_d, N_samples=args.N_samples, near=ds.near, far=ds.far)
            R = rays_o.shape[0]
            ret = model_forward(pts=pts, dirs=rays_d.unsqueeze(1).expand([R, args.N_samples, 3]))
            _, _, weights = utils.alpha_composition(ret['rgb'], ret['sigma'], ts, 0)
            pts_f, ts_f = samp_func_fine(rays_o, rays_d, ts, weights, args.N_samples_fine)
            n = args.N_samples + args.N_samples_fine
            ret = model_forward_fine(pts=pts_f, dirs=rays_d.unsqueeze(1).expand([R, n, 3]))
            rgb_f, t_f, _ = utils.alpha_composition(ret['rgb'], ret['sigma'], ts_f, 0)
        rgb_np, t_np = rgb_f.detach().cpu().numpy(), t_f.detach().cpu().numpy()
        coor = t_np[..., None] * rays_d.detach().cpu().numpy() + rays_o.detach().cpu().numpy()   # rendering.py:54
        n = coor.shape[0]
        rgb_map[pixel_id:pixel_id + n], t_map[pixel_id:pixel_id + n], coor_map[pixel_id:pixel_id + n] = rgb_np, t_np, coor
        pixel_id += n
        done = pixel_id // res - img_id
        if done > 0 and sv_path is not None:
            for i in range(img_id, img_id + done):
                sl = slice(i * res, (i + 1) * res)
                gid = _image_id(ds, i)
                _write_depth_rgb(sv_path, rgb_map[sl], t_map[sl], h, w, 'rgb_%05d.png' % gid, 'depth_%05d.png' % gid)
                np.savez(os.path.join(sv_path, 'geometry_%05d' % gid), coor_map=coor_map[sl].reshape(h, w, 3),
                         cps=cps[gid], hwf=ds.hwf, near=ds.near, far=ds.far)
        img_id += max(done, 0)
    rgb_map, t_map = rgb_map.reshape(-1, h, w, 3), t_map.reshape(-1, h, w, 1)
    _drain_images()
    if sv_path is not None:
        if world > 1:
            # the scene-wide file: rank 0 puts the per-frame files of all ranks together (one node, one file system)
            ds.dist.barrier()
            if rank == 0:
                whole = np.stack([np.load(os.path.join(sv_path, 'geometry_%05d.npz' % k))['coor_map'] for k in range(frame_num)])
                np.savez(os.path.join(sv_path, 'geometry'), coor_map=whole, cps=cps, hwf=ds.hwf, near=ds.near, far=ds.far)
            ds.dist.barrier()
        else:
            np.savez(os.path.join(sv_path, 'geometry'), coor_map=coor_map.reshape(-1, h, w, 3), cps=cps, hwf=ds.hwf,
                     near=ds.near, far=ds.far)
    return rgb_map, t_map


def _styled_batch(b, args, ds, samp_func, model_forward, style_forward, concat_style_forward, latents_model_1,
                  model_forward_fine, samp_func_fine, renderer):
    """One batch of the stylised chain (rendering.py:118-178 == :280-327)."""
    from . import utils
    rays_o, rays_d = b['rays_o'], b['rays_d']
    z = latents_model_1(style_ids=b['style_id'].long(), frame_ids=b['frame_id'].long(), type=args.dataset_type)
    if renderer is not None:
        # stratified jitter (utils.py:518-524, perturb=True at rendering.py:118,280): the reference draws it per batch;
        # a dataset may deliver it per ray instead, so that the image does not depend on batching or sharding
        jitter = b['jitter'] if 'jitter' in b else torch.rand(rays_o.shape[0], args.N_samples, device=rays_o.device)
        out = renderer.render(rays_o, rays_d, args.N_samples, args.N_samples_fine, near=ds.near, far=ds.far,
                              jitter=jitter, z=z)
        return out["rgb"], out["t"]
    R, L = rays_o.shape[0], z.shape[-1]
    zbar = torch.mean(z, dim=1, keepdim=True)                      # rendering.py:126

    def one_pass(fwd, pts, n):
        ret = fwd(pts=pts, dirs=rays_d.unsqueeze(1).expand([R, n, 3]))
        cf = concat_style_forward(x=ret['pts'], latent=z.unsqueeze(1).expand([R, n, L]))['concat_features']
        both = torch.cat((ret['base_remap'], cf), dim=-1)           # rendering.py:132
        rgb = style_forward(x=ret['pts'], concated=both, latent=zbar.unsqueeze(2).expand([R, n, L]))['rgb']
        return rgb, ret['sigma']

    pts, ts = samp_func(rays_o=rays_o, rays_d=rays_d, N_samples=args.N_samples, near=ds.near, far=ds.far, perturb=True)
    rgb, sig = one_pass(model_forward, pts, args.N_samples)
    _, _, weights = utils.alpha_composition(rgb, sig, ts, 0)
    pts_f, ts_f = samp_func_fine(rays_o, rays_d, ts, weights, args.N_samples_fine)
    rgb, sig = one_pass(model_forward_fine, pts_f, args.N_samples + args.N_samples_fine)
    rgb_f, t_f, _ = utils.alpha_composition(rgb, sig, ts_f, 0)
    return rgb_f, t_f


def render_style(model_forward, samp_func, style_forward, concat_style_forward, latents_model_1, dataloader, args,
                 device, sv_path=None, model_forward_fine=None, samp_func_fine=None, sigma_scale=0., renderer=None):
    """reference rendering.py:93-239: stylised render of the `valid_style` rays; one
    style_%05d_fine_%05d.png + style_%05d_fine_depth_%05d.png pair per completed frame.
    Returns (rgb_map_fine, t_map_fine) = the rays left over after the last whole image, like the reference."""
    _require_fine(args)
    latents_model_1.sigma_scale = sigma_scale
    if sv_path is not None:
        os.makedirs(sv_path, exist_ok=True)
    ds = dataloader.dataset
    ds.mode = 'valid_style'
    frame_num, h, w = ds.cps_valid.shape[0], ds.h, ds.w
    res = _local_res(ds, h * w)
    pend_rgb, pend_t, image_no = torch.zeros([0, 3], device=device), torch.zeros([0], device=device), 0
    for batch in dataloader:
        b = _to_device(batch, device)
        rgb_f, t_f = _styled_batch(b, args, ds, samp_func, model_forward, style_forward, concat_style_forward,
                                   latents_model_1, model_forward_fine, samp_func_fine, renderer)
        pend_rgb = torch.cat([pend_rgb, rgb_f.detach().float()], 0)      # stays on the device until a frame is complete
        pend_t = torch.cat([pend_t, t_f.detach().float()], 0)
        while pend_rgb.shape[0] >= res:
            rgb_img, t_img, writer = _assemble(ds, pend_rgb[:res], pend_t[:res])
            if sv_path is not None and writer:
                # file numbering: images are consecutive (style, frame) pairs (rendering.py:209-218)
                gid = _image_id(ds, image_no)
                _write_depth_rgb(sv_path, rgb_img, t_img, h, w,
                                 'style_%05d_fine_%05d.png' % (gid // frame_num, gid % frame_num),
                                 'style_%05d_fine_depth_%05d.png' % (gid // frame_num, gid % frame_num))
            image_no += 1
            pend_rgb, pend_t = pend_rgb[res:], pend_t[res:]
    _drain_images()
    return pend_rgb.cpu().numpy(), pend_t.cpu().numpy()


def render_train_style(samp_func, model_forward, style_forward, concat_style_forward, latents_model_1, dataset, args,
                       device, sv_path=None, model_forward_fine=None, samp_func_fine=None, sigma_scale=0.,
                       renderer=None):
    """reference rendering.py:242-375: stylised render of the training views in `train_style` order; the batch is the
    largest divisor of h*w not above --chunk (:251-253); images already on disk are skipped (:267-270); RGB is clamped
    to [0,1] (:328); depth is min-max normalised without epsilon and written as 3 channels (:358-361)."""
    _require_fine(args)
    os.makedirs(sv_path, exist_ok=True)
    latents_model_1.sigma_scale = sigma_scale
    frame_num, h, w = dataset.frame_num, dataset.h, dataset.w
    dataset.mode = 'train_style'
    batch_size = args.chunk
    local = _local_res(dataset, h * w)
    while local % batch_size != 0:
        batch_size -= 1
    iters_per_image = local // batch_size
    loader = dataset.batches(batch_size) if hasattr(dataset, 'batches') else torch.utils.data.DataLoader(
        dataset, shuffle=False, batch_size=batch_size, num_workers=getattr(args, 'num_workers', 0))
    it = img_count = 0
    rgbs, ts = [], []
    for batch in loader:
        gid = _image_id(dataset, img_count)
        path = os.path.join(sv_path, 'style_%05d_fine_%05d.png' % (gid // frame_num, gid % frame_num))
        # (rays sharding: every rank must take the same decision, or the all-gather of a frame would hang)
        exists = os.path.exists(path) and not (getattr(dataset, 'world', 1) > 1 and getattr(dataset, 'shard', '') == 'rays')
        if not exists:
            b = _to_device(batch, device)
            rgb_f, t_f = _styled_batch(b, args, dataset, samp_func, model_forward, style_forward, concat_style_forward,
                                       latents_model_1, model_forward_fine, samp_func_fine, renderer)
            rgbs.append(torch.clamp(rgb_f, 0., 1.).detach())
            ts.append(t_f.detach())
        it += 1
        if it == iters_per_image:
            if not exists:
                rgb_img, t_img, writer = _assemble(dataset, torch.cat(rgbs, 0), torch.cat(ts, 0))
                if writer:
                    _write_depth_rgb(sv_path, rgb_img, t_img, h, w, os.path.basename(path),
                                     os.path.basename(path).replace('_fine_', '_fine_depth_'), eps=0., depth_channels=3)
            img_count += 1
            it, rgbs, ts = 0, [], []
    _drain_images()
    return img_count
