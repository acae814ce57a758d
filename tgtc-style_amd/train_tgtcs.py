"""`python -m tgtc_style_amd.train_tgtcs --config configs/fern.txt --render_valid_style [--synthetic]`

The render half of the reference entry point (train_tgtcs.py:13-215): build the four networks and the latent
table, reload the newest checkpoints, and dispatch to render_style / render_train_style (or cal_geometry with
--render_valid).  Output directory and file names are the reference's (train_tgtcs.py:20,164,196;
rendering.py:216-217,363-364).  Training loops are out of scope.

No dataset or checkpoint ships with this repository (and the LLFF loader is a `next` row of SURVEY section 8f), so
`--synthetic` provides a seeded scene: synth.nerf_state/... weights and a closed-form camera path.  Without it
the reference layout is expected: <sv_path>/NNNNNN.tar, style_NNNNNN.tar, latent_NNNNNN.tar.
"""
import os
import sys

import numpy as np
import torch

from . import config as cfg
from . import checkpoints, models, rendering, synth, utils


class ShardedScene:
    """Multi-GPU side of the dataset duck type (one process per GPU, torch.distributed).

    shard = 'frames' (BASELINE config 5): rank r owns the images k with k % world == r, renders them whole and writes
    their files itself -- no collective at all.  shard = 'rays' (config 4): every rank renders the contiguous pixel
    range parallel.shard_range(h*w, rank, world) of EVERY image; a finished image is reassembled by one all-gather
    of [rays, 4] rows (RCCL over xGMI with the nccl backend) and written by rank 0."""
    rank, world, shard, dist = 0, 1, 'frames', None
    jitter_samples, jitter_seed = 0, 0     # > 0: batches carry per-ray stratified jitter drawn from a per-IMAGE generator

    def image_jitter(self, k):
        """[h*w, N_samples] uniforms of image k: the same rows whatever the batch size, rank or sharding."""
        g = torch.Generator(device=self.device)
        g.manual_seed(self.jitter_seed + 1000003 * k)
        return torch.rand(self.h * self.w, self.jitter_samples, generator=g, device=self.device)

    def set_sharding(self, rank, world, shard, dist=None):
        self.rank, self.world, self.shard, self.dist = rank, world, shard, dist

    def _images(self, n_img):
        return [k for k in range(n_img) if self.shard == 'rays' or k % self.world == self.rank]

    def _pixels(self):
        from . import parallel
        return parallel.shard_range(self.h * self.w, self.rank, self.world) if self.shard == 'rays' else (0, self.h * self.w)

    # hooks read by rendering.py
    def global_image(self, local_no):
        return local_no if self.shard == 'rays' else local_no * self.world + self.rank

    def rays_per_image(self):
        lo, hi = self._pixels()
        return hi - lo

    def assemble(self, rgb, t):
        if self.world == 1 or self.shard != 'rays':
            return rgb, t, True
        from . import parallel
        rows = parallel.gather_rows(torch.cat([rgb, t[:, None]], 1).contiguous(), self.h * self.w, self.rank, self.world, self.dist)
        return rows[:, :3], rows[:, 3], self.rank == 0


class SyntheticScene(ShardedScene):
    """Duck type of the reference datasets for the render drivers (dataset.py:361-470 attributes), with rays
    generated on the device per frame instead of stored float64 tables (dataset.py:412-433)."""

    def __init__(self, h, w, frames, valid_frames, device="cuda", style_num=1):
        self.h, self.w, self.f = h, w, synth.fern_intrinsics(h, w)
        self.hwf = [h, w, self.f]
        self.near, self.far = 0., 1.
        to44 = lambda p: np.concatenate([p, np.array([[0, 0, 0, 1]], np.float32)], 0)
        self.cps = np.stack([to44(synth.spiral_pose(7 * i)) for i in range(frames)])
        self.cps_valid = np.stack([to44(synth.spiral_pose(i, n=valid_frames)) for i in range(valid_frames)])
        self.frame_num, self.style_num = frames, style_num
        self.mode, self.device = 'train', device

    def batches(self, batch_size):
        poses = self.cps if 'train' in self.mode else self.cps_valid
        n_img = len(poses) * (self.style_num if 'style' in self.mode else 1)
        first, last = self._pixels()
        for k in self._images(n_img):
            sid, fid = divmod(k, len(poses))
            o, d = utils.gen_rays(self.h, self.w, self.f, poses[fid][:3, :4], first_pixel=first, n=last - first,
                                  device=self.device)
            jit = self.image_jitter(k)[first:last] if self.jitter_samples and 'style' in self.mode else None
            for lo in range(0, last - first, batch_size):
                n = min(batch_size, last - first - lo)
                batch = {'rays_o': o[lo:lo + n], 'rays_d': d[lo:lo + n],
                         'style_id': torch.full((n,), sid, dtype=torch.long), 'frame_id': torch.full((n,), fid, dtype=torch.long)}
                if jit is not None:
                    batch['jitter'] = jit[lo:lo + n]
                yield batch


class LlffPoseScene(SyntheticScene):
    """A real LLFF scene as far as rendering needs it: camera poses from `<datadir>/poses_bounds.npy`
    (llff_poses.scene_poses = load_llff.load_llff_data without the images; dataset.py:69-102), rays generated on the
    device.  Training views = the recentred poses, validation views = the 120-view spiral (`valid_frames` of it)."""

    def __init__(self, datadir, factor, device="cuda", valid_frames=None, style_num=1):
        from . import llff_poses
        arr = np.load(os.path.join(datadir, 'poses_bounds.npy'))
        h0, w0 = arr[0, :15].reshape(3, 5)[:2, 4]
        sc = llff_poses.scene_poses(arr, (int(h0 // factor), int(w0 // factor)), factor=factor)
        self.h, self.w, self.f = int(sc["hwf"][0]), int(sc["hwf"][1]), float(sc["hwf"][2])
        self.hwf = [self.h, self.w, self.f]
        self.near, self.far = 0., 1.                                  # NDC (dataset.py:379-380)
        self.cps = llff_poses.valid_camera_poses(sc["poses"])
        self.cps_valid = llff_poses.valid_camera_poses(sc["render_poses"])[:valid_frames]
        self.frame_num, self.style_num = self.cps.shape[0], style_num
        self.mode, self.device = 'train', device


class _Loader:
    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size

    def __iter__(self):
        return self.dataset.batches(self.batch_size)


def _t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _init_distributed():
    """One process per GPU under torchrun (RANK / WORLD_SIZE / LOCAL_RANK); the process group comes up BEFORE the first
    HIP call of the process.  TGTC_DIST_BACKEND=gloo lets several ranks share a GPU (functional rehearsals, tests)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, 0, None
    import torch.distributed as dist
    backend = os.environ.get("TGTC_DIST_BACKEND", "nccl")
    if dist.is_initialized():          # a batch of scenes in one process (render_batch.py): the group is already up
        return rank, world, (local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)), dist
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
        local = local % max(torch.cuda.device_count(), 1)
    return rank, world, local, dist


def _broadcast_from_rank0(t, dist):
    """In-place broadcast of a parameter table; gloo (several test ranks on one GPU) moves it through host memory."""
    if dist.get_backend() == "nccl":
        dist.broadcast(t, 0)
    else:
        host = t.detach().cpu()
        dist.broadcast(host, 0)
        t.copy_(host)


def train(args):
    rank, world, local, dist = _init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("train_tgtcs: no GPU visible; the HIP render path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_viewdir_str = '_UseViewDir_' if args.use_viewdir else ''
    sv_path = os.path.join(args.basedir, args.expname + '_' + args.nerf_type + '_' + args.act_type + use_viewdir_str +
                           'ImgFactor' + str(int(args.factor)))        # train_tgtcs.py:19-20
    os.makedirs(sv_path, exist_ok=True)

    model = models.StyleNerf(args, mode='coarse').to(device)
    model_fine = models.StyleNerf(args, mode='fine').to(device)
    concat_model = models.StyleMLP_before_concat(args).to(device)
    style_model = models.StyleMLP_Wild_multilayers(args).to(device)
    global_step = 0
    step = None if args.no_reload else checkpoints.load_nerf(sv_path, model, model_fine)           # train_tgtcs.py:60-72
    if step is not None:
        global_step = step
    elif args.synthetic:
        model.load_state_dict(_t(synth.nerf_state(0)))
        model_fine.load_state_dict(_t(synth.nerf_state(1)))
    else:
        raise SystemExit("train_tgtcs: no NeRF checkpoint in %s (use --synthetic for the seeded scene)" % sv_path)
    step = None if args.no_reload else checkpoints.load_style(sv_path, style_model, concat_model)   # train_tgtcs.py:74-82
    if step is not None:
        global_step = step
    elif args.synthetic:
        concat_model.load_state_dict(_t(synth.concat_state(2)))
        style_model.load_state_dict(_t(synth.style_state(3)))

    # what the 2-D pass left next to the scene (dataset.py:437-440): style images, their 1024-d features, their number
    from . import trans_test
    stylized = trans_test.read_stylized_data(args.datadir, args.factor)
    if os.path.exists(os.path.join(args.datadir, 'poses_bounds.npy')):
        # rendering needs the cameras of the scene, not its images
        dataset = LlffPoseScene(args.datadir, args.factor, device=device,
                                valid_frames=args.synthetic_frames if args.synthetic else None,
                                style_num=stylized["style_num"] if stylized else 1)
    elif args.synthetic:
        hw = args.synthetic_hw
        dataset = SyntheticScene(hw, hw, frames=20, valid_frames=args.synthetic_frames, device=device)
    else:
        raise SystemExit("train_tgtcs: no poses_bounds.npy in --datadir %s (the image side of the LLFF loader is not part "
                         "of this build, SURVEY section 8f); run with --synthetic for the seeded scene" % args.datadir)
    dataset.jitter_samples = args.N_samples     # per-ray jitter: images independent of --batch_size / --chunk / sharding
    if world > 1:
        if (args.render_valid or args.render_train) and args.shard != 'frames':
            raise SystemExit("train_tgtcs: the geometry pass (--render_valid / --render_train) shards by whole frames "
                             "(--shard frames): geometry_%05d.npz is a per-frame file")
        dataset.set_sharding(rank, world, args.shard, dist)
    # rays per render call: the reference feeds --batch_size rays at a time from its host loader; with device-generated rays
    # a whole image (this rank's part of it) per call keeps the persistent kernels busy -- 78 calls of 2 048 rays cost a
    # 400x400 frame ~30 ms of launch gaps -- and gives the same pixels
    batch_size = args.batch_size if args.literal_batches else max(args.batch_size, dataset.rays_per_image())
    latents = models.StyleLatents_variational(style_num=dataset.style_num, frame_num=dataset.frame_num,
                                              latent_dim=args.vae_latent).to(device)
    if not args.no_reload and checkpoints.load_latents(sv_path, latents):                  # train_tgtcs.py:139-146
        pass
    elif stylized is not None and dataset.style_num == stylized["style_num"] and os.path.exists(args.vae_pth_path):
        # train_tgtcs.py:128-155: no latent checkpoint -> the VAE encodes the style features into mu / logvar and every
        # frame's latent is drawn around them
        vae = models.VAE(data_dim=1024, latent_dim=args.vae_latent, W=args.vae_w, D=args.vae_d, kl_lambda=args.vae_kl_lambda)
        vae.load_state_dict(torch.load(args.vae_pth_path, map_location='cpu'))
        vae.eval().to(device)
        feats = torch.from_numpy(np.asarray(stylized["style_features"], np.float32)).to(device)
        _, mu, logvar = vae.encode(feats)
        latents.style_latents_mu = torch.nn.Parameter(mu.detach())
        latents.style_latents_logvar = torch.nn.Parameter(logvar.detach())
        latents.set_latents(generator=torch.Generator().manual_seed(args.latent_seed) if args.latent_seed >= 0 else None)
        if world > 1:
            # the draw comes from each process's own unseeded generator: rank 0's table is THE table, or the stripes / frames
            # of different ranks would be rendered with different style latents
            latents = latents.to(device)
            for p in (latents.latents, latents.style_latents_mu, latents.style_latents_logvar):
                _broadcast_from_rank0(p.data, dist)
        print('Initializing Latent Model from', args.vae_pth_path)
    elif args.synthetic:
        latents.load_state_dict(_t(synth.latents_state(4, style_num=dataset.style_num, frame_num=dataset.frame_num)))
    else:
        raise SystemExit("train_tgtcs: no latent checkpoint in %s and no stylized_data.npz + --vae_pth_path to initialise "
                         "the latents from (use --synthetic for the seeded scene)" % sv_path)
    latents = latents.to(device)

    renderer = rendering.RayRenderer(model, model_fine, models.StylePair(concat_model, style_model))
    common = dict(samp_func=utils.sampling_pts_uniform, model_forward=utils.batchify(lambda **kw: model(**kw), args.chunk),
                  model_forward_fine=utils.batchify(lambda **kw: model_fine(**kw), args.chunk),
                  samp_func_fine=utils.sampling_pts_fine_torch, args=args, device=device)
    styled = dict(style_forward=utils.batchify(lambda **kw: style_model(**kw), args.chunk),
                  concat_style_forward=utils.batchify(lambda **kw: concat_model(**kw), args.chunk),
                  latents_model_1=latents, sigma_scale=args.sigma_scale, renderer=renderer)
    with torch.no_grad():
        if args.render_valid_style:
            out = os.path.join(sv_path, 'render_valid_' + str(global_step))
            model.set_enable_style(True), model_fine.set_enable_style(True)
            dataset.mode = 'valid_style'
            rendering.render_style(dataloader=_Loader(dataset, batch_size), sv_path=out, **common, **styled)
            print('Done, saving to', out)
            return out
        if args.render_train_style:
            out = os.path.join(sv_path, 'render_train_' + str(global_step))
            model.set_enable_style(True), model_fine.set_enable_style(True)
            rendering.render_train_style(dataset=dataset, sv_path=out, **common, **styled)
            print('Done, saving to', out)
            return out
        if args.render_valid or args.render_train:
            out = os.path.join(sv_path, 'nerf_gen_data2')
            dataset.mode = 'train' if args.render_train else 'valid'
            rendering.cal_geometry(dataloader=_Loader(dataset, batch_size), sv_path=out,
                                   renderer=rendering.RayRenderer(model, model_fine), **common)
            print('Done, saving to', out)
            return out
    raise SystemExit("train_tgtcs: nothing to do -- pass --render_valid_style, --render_train_style or --render_valid "
                     "(the training loops of the reference are outside this build)")


def _finish_distributed():
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def main(argv=None):
    args = cfg.parse_args(argv)
    if args.expname is None:
        raise SystemExit("train_tgtcs: --expname (or --config) is required")
    try:
        return train(args)      # the reference wraps this in `while True` (train_tgtcs.py:596-597); once is enough
    finally:
        _finish_distributed()


if __name__ == '__main__':
    main()
