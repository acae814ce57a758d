"""GPU: the train_tgtcs-compatible CLI end to end on the synthetic scene (BASELINE config 1 "plumbing", but on the
HIP path): image files with the reference's names appear, and the drop-in drivers fed with granular HIP callables
agree with the fused renderer."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_render_valid_style_and_train_style(tmp_path):
    from PIL import Image
    from tgtc_style_amd import train_tgtcs
    base = ["--config", os.path.join(ROOT, "configs", "fern.txt"), "--basedir", str(tmp_path), "--synthetic",
            "--synthetic_hw", "32", "--synthetic_frames", "2", "--chunk", "1024", "--batch_size", "512"]
    out = train_tgtcs.main(base + ["--render_valid_style"])
    # train_tgtcs.py:20,164: basedir/expname_nerftype_act_UseViewDir_ImgFactorN/render_valid_<step>
    assert out == os.path.join(str(tmp_path), "fern_style_style_nerf_relu_UseViewDir_ImgFactor4", "render_valid_0")
    names = sorted(os.listdir(out))
    assert names == ["style_00000_fine_00000.png", "style_00000_fine_00001.png",
                     "style_00000_fine_depth_00000.png", "style_00000_fine_depth_00001.png"]      # rendering.py:216-217
    img = np.asarray(Image.open(os.path.join(out, names[0])))
    assert img.shape == (32, 32, 3) and img.dtype == np.uint8 and img.std() > 0
    out = train_tgtcs.main(base + ["--render_train_style", "--chunk", "300"])
    names = sorted(os.listdir(out))
    assert len(names) == 40 and names[0] == "style_00000_fine_00000.png" and names[-1] == "style_00000_fine_depth_00019.png"
    # resume-by-skipping (rendering.py:267-270): a second run leaves the files untouched
    stamp = os.path.getmtime(os.path.join(out, names[0]))
    train_tgtcs.main(base + ["--render_train_style", "--chunk", "300"])
    assert os.path.getmtime(os.path.join(out, names[0])) == stamp
    with pytest.raises(SystemExit):
        train_tgtcs.main(base)        # nothing to do


def test_cal_geometry_dropin_matches_fused(tmp_path):
    """cal_geometry with the reference's injected-callable signature (granular HIP operators through batchify) vs the
    fused single-call renderer: same images, same geometry files."""
    from tgtc_style_amd import config as cfg, models, rendering, synth, train_tgtcs, utils
    args = cfg.parse_args(["--config", os.path.join(ROOT, "configs", "fern.txt"), "--N_samples", "128", "--chunk", "200"])
    t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    nets = []
    for seed, mode in ((0, "coarse"), (1, "fine")):
        m = models.StyleNerf(args, mode=mode)
        m.load_state_dict(t(synth.nerf_state(seed)))
        nets.append(m.cuda())
    ds = train_tgtcs.SyntheticScene(16, 24, frames=2, valid_frames=2)
    kw = dict(samp_func=utils.sampling_pts_uniform, samp_func_fine=utils.sampling_pts_fine_torch, args=args, device="cuda",
              model_forward=utils.batchify(lambda **k: nets[0](**k), args.chunk),
              model_forward_fine=utils.batchify(lambda **k: nets[1](**k), args.chunk))
    a_rgb, a_t = rendering.cal_geometry(dataloader=train_tgtcs._Loader(ds, 100), sv_path=str(tmp_path / "a"), **kw)
    b_rgb, b_t = rendering.cal_geometry(dataloader=train_tgtcs._Loader(ds, 384), sv_path=str(tmp_path / "b"),
                                        renderer=rendering.RayRenderer(*nets), **kw)
    assert a_rgb.shape == (2, 16, 24, 3) and a_t.shape == (2, 16, 24, 1)
    # both routes run the same kernels on the same rays; only the coarse pass differs (full vs sigma-only kernel)
    assert np.abs(a_rgb - b_rgb).max() <= 1e-5 and np.abs(a_t - b_t).max() <= 1e-5
    assert sorted(os.listdir(tmp_path / "a")) == ["depth_00000.png", "depth_00001.png", "geometry.npz",
                                                  "geometry_00000.npz", "geometry_00001.npz", "rgb_00000.png", "rgb_00001.png"]
    geo = np.load(tmp_path / "a" / "geometry_00001.npz")
    assert sorted(geo.files) == ["coor_map", "cps", "far", "hwf", "near"] and geo["coor_map"].shape == (16, 24, 3)


def test_cli_renders_llff_camera_path(tmp_path, golden):
    """A scene given by its poses_bounds.npy: the CLI renders the reference's spiral validation path
    (llff_poses.scene_poses, pinned to load_llff_data by golden g11) with rays generated on the device."""
    from PIL import Image
    from tgtc_style_amd import llff_poses, train_tgtcs, utils
    g = golden("g11_llff_poses")
    scene = tmp_path / "scene"
    scene.mkdir()
    np.save(scene / "poses_bounds.npy", g["poses_arr"])
    out = train_tgtcs.main(["--config", os.path.join(ROOT, "configs", "fern.txt"), "--basedir", str(tmp_path), "--datadir", str(scene),
                            "--factor", "8", "--synthetic", "--synthetic_frames", "3", "--chunk", "1024", "--batch_size", "100",
                            "--render_valid_style"])
    names = sorted(os.listdir(out))
    assert names == ["style_00000_fine_%05d.png" % i for i in range(3)] + ["style_00000_fine_depth_%05d.png" % i for i in range(3)]
    img = np.asarray(Image.open(os.path.join(out, names[1])))
    assert img.shape == (12, 16, 3) and img.std() > 0            # 96x128 scene at factor 8
    # the rays the driver used for frame 1 are the reference's: golden cps_valid through the (golden-checked) ray generator
    ds = train_tgtcs.LlffPoseScene(str(scene), 8, valid_frames=3)
    assert np.abs(ds.cps_valid - g["cps_valid"][:3]).max() <= 2e-6 * np.abs(g["cps_valid"]).max() and (ds.h, ds.w) == (12, 16)
    ds.mode = 'valid_style'
    batches = list(ds.batches(1000))
    o, d = utils.gen_rays(12, 16, float(g["render_poses"][1, 2, 4]), g["cps_valid"][1, :3, :4])
    assert torch.allclose(batches[1]['rays_o'], o, atol=1e-6) and torch.allclose(batches[1]['rays_d'], d, atol=1e-6)


def test_cli_reloads_reference_checkpoints(tmp_path):
    """Checkpoints in the reference's on-disk layout (train_tgtcs.py:285-300 `NNNNNN.tar`, :504-517 `style_NNNNNN.tar` /
    `latent_NNNNNN.tar`; loaded at :60-82, :139-146): the newest of each kind is picked up, `global_step` names the
    output directory, and the images are the ones these weights render (not the synthetic defaults)."""
    from PIL import Image
    from tgtc_style_amd import synth, train_tgtcs
    t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    base = ["--config", os.path.join(ROOT, "configs", "fern.txt"), "--basedir", str(tmp_path), "--synthetic",
            "--synthetic_hw", "16", "--synthetic_frames", "1", "--chunk", "1024", "--batch_size", "256", "--render_valid_style"]
    ref_dir = train_tgtcs.main(base)                                   # synthetic weights, step 0
    sv = os.path.dirname(ref_dir)
    torch.save({"global_step": 100, "model": t(synth.nerf_state(10)), "model_fine": t(synth.nerf_state(11)), "optimizer": {}},
               os.path.join(sv, "000100.tar"))
    torch.save({"global_step": 120000, "model": t(synth.nerf_state(0)), "model_fine": t(synth.nerf_state(1)), "optimizer": {},
                "style_optimizer": {}}, os.path.join(sv, "120000.tar"))                        # the newest NeRF checkpoint wins
    torch.save({"global_step": 120500, "model": t(synth.style_state(13)), "concat_model": t(synth.concat_state(12)), "optimizer": {}},
               os.path.join(sv, "style_120500.tar"))
    torch.save({"global_step": 120500, "train_set_1": t(synth.latents_state(14, style_num=1, frame_num=20))},
               os.path.join(sv, "latent_120500.tar"))
    out = train_tgtcs.main(base)
    assert os.path.basename(out) == "render_valid_120500"              # the style checkpoint's global_step (train_tgtcs.py:79)
    a = np.asarray(Image.open(os.path.join(out, "style_00000_fine_00000.png"))).astype(int)
    b = np.asarray(Image.open(os.path.join(ref_dir, "style_00000_fine_00000.png"))).astype(int)
    assert np.abs(a - b).max() > 8                                     # other style / latent weights: a different image
    # --no_reload ignores the files again (train_tgtcs.py:63)
    assert os.path.basename(train_tgtcs.main(base + ["--no_reload"])) == "render_valid_0"


def _cli_rank(rank, world, port, argv, backend_env):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), TGTC_DIST_BACKEND=backend_env)
    from tgtc_style_amd import train_tgtcs
    train_tgtcs.main(argv)


@pytest.mark.parametrize("shard", ["frames", "rays"])
def test_cli_two_ranks_write_the_single_rank_images(tmp_path, shard):
    """The CLI under torchrun-style environment variables (two ranks, gloo rendezvous on 127.0.0.1, both on the test
    box's one GPU): --shard frames = images round-robin with rank-local files, --shard rays = contiguous ray ranges of
    every image + one all-gather, rank 0 writes.  Either way the PNG files are byte-identical to the one-rank run."""
    import socket
    import torch.multiprocessing as mp
    from tgtc_style_amd import train_tgtcs
    common = ["--config", os.path.join(ROOT, "configs", "fern.txt"), "--synthetic", "--synthetic_hw", "20", "--synthetic_frames", "3",
              "--chunk", "1024", "--batch_size", "128", "--render_valid_style"]
    one = train_tgtcs.main(common + ["--basedir", str(tmp_path / "one")])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TGTC_DIST_BACKEND")}
    try:
        mp.spawn(_cli_rank, args=(2, port, common + ["--basedir", str(tmp_path / "two"), "--shard", shard], "gloo"), nprocs=2, join=True)
    finally:
        for k, v in env.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    two = os.path.join(str(tmp_path / "two"), os.path.relpath(one, str(tmp_path / "one")))
    names = sorted(os.listdir(one))
    assert len(names) == 6 and sorted(os.listdir(two)) == names
    for n in names:
        with open(os.path.join(one, n), "rb") as a, open(os.path.join(two, n), "rb") as b:
            assert a.read() == b.read(), n


def test_vae_latent_initialisation(tmp_path, golden):
    """No latent checkpoint (train_tgtcs.py:128-155): the VAE encodes the style features left by the 2-D pass into the
    latent table's mu / logvar, every frame's latent is drawn around them.  VAE.encode on the HIP GEMM vs the reference's
    own outputs (golden g12); then the CLI on a scene directory in the reference's layout -- poses_bounds.npy,
    stylized_gen_<factor>/stylized_data.npz, vae.pth, NNNNNN.tar, style_NNNNNN.tar -- without --synthetic."""
    from tgtc_style_amd import models, synth, train_tgtcs
    t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    g = golden("g12_vae")
    vae = models.VAE(data_dim=1024, latent_dim=32, W=512, D=4)
    vae.load_state_dict(t(synth.vae_state(9)))
    vae.cuda()
    z, mu, logvar = vae.encode(torch.from_numpy(g["style_features"]).cuda())
    assert z is mu
    assert float((mu.cpu() - torch.from_numpy(g["mu"])).abs().max()) <= 2e-5 * float(np.abs(g["mu"]).max())
    assert float((logvar.cpu() - torch.from_numpy(g["logvar"])).abs().max()) <= 2e-5 * float(np.abs(g["logvar"]).max())
    lat = models.StyleLatents_variational(style_num=3, frame_num=5, latent_dim=32).cuda()
    lat.style_latents_mu, lat.style_latents_logvar = torch.nn.Parameter(mu), torch.nn.Parameter(logvar)
    lat.set_latents(generator=torch.Generator().manual_seed(0))
    assert tuple(lat.latents.shape) == tuple(g["latents_shape"]) and lat.latents.is_cuda
    dev = (lat.latents.detach() - mu[:, None, :]) / torch.exp(0.5 * logvar)[:, None, :]          # ~ N(0,1)
    assert abs(float(dev.mean())) < 0.2 and 0.8 < float(dev.std()) < 1.2

    scene = tmp_path / "scene"
    (scene / "stylized_gen_8.0").mkdir(parents=True)
    np.save(scene / "poses_bounds.npy", golden("g11_llff_poses")["poses_arr"])
    np.savez(scene / "stylized_gen_8.0" / "stylized_data", style_names={"s": 0}, style_paths="style/s.jpg",
             style_images=np.zeros([1, 8, 8, 3], np.float32), style_features=g["style_features"][:1])
    torch.save(t(synth.vae_state(9)), tmp_path / "vae.pth")
    argv = ["--config", os.path.join(ROOT, "configs", "fern.txt"), "--basedir", str(tmp_path), "--datadir", str(scene), "--factor", "8",
            "--vae_pth_path", str(tmp_path / "vae.pth"), "--chunk", "1024", "--batch_size", "100", "--render_train_style"]
    with pytest.raises(SystemExit, match="no NeRF checkpoint"):
        train_tgtcs.main(argv)
    sv = os.path.join(str(tmp_path), "fern_style_style_nerf_relu_UseViewDir_ImgFactor8")
    torch.save({"global_step": 7, "model": t(synth.nerf_state(0)), "model_fine": t(synth.nerf_state(1))}, os.path.join(sv, "000007.tar"))
    torch.save({"global_step": 9, "model": t(synth.style_state(3)), "concat_model": t(synth.concat_state(2))},
               os.path.join(sv, "style_000009.tar"))
    out = train_tgtcs.main(argv)
    names = sorted(os.listdir(out))
    assert os.path.basename(out) == "render_train_9" and len(names) == 2 * 20 and names[0] == "style_00000_fine_00000.png"
    # without the VAE file there is nothing to initialise the latents from
    with pytest.raises(SystemExit, match="no latent checkpoint"):
        train_tgtcs.main(argv[:argv.index("--vae_pth_path")] + ["--vae_pth_path", str(tmp_path / "missing.pth")] + argv[argv.index("--vae_pth_path") + 2:])


def _cli_rank_seeded(rank, world, port, argv, backend_env):
    torch.manual_seed(1234 + rank)      # every process has its own default generator, like separately started ranks
    _cli_rank(rank, world, port, argv, backend_env)


@pytest.mark.parametrize("shard", ["rays", "frames"])
def test_vae_latents_are_one_table_across_ranks(tmp_path, golden, shard):
    """The VAE-initialised latent table is drawn from the process's unseeded default generator (train_tgtcs.py:128-155).
    Under torchrun rank 0's draw is broadcast, so two ranks (whose generators differ) write the files of the one-rank
    run whose generator equals rank 0's -- no seams between the stripes of `--shard rays`."""
    import socket
    import torch.multiprocessing as mp
    from tgtc_style_amd import synth, train_tgtcs
    t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    g = golden("g12_vae")
    scene = tmp_path / "scene"
    (scene / "stylized_gen_8.0").mkdir(parents=True)
    np.save(scene / "poses_bounds.npy", golden("g11_llff_poses")["poses_arr"])
    np.savez(scene / "stylized_gen_8.0" / "stylized_data", style_names={"s": 0}, style_paths="style/s.jpg",
             style_images=np.zeros([1, 8, 8, 3], np.float32), style_features=g["style_features"][:1])
    torch.save(t(synth.vae_state(9)), tmp_path / "vae.pth")
    outs = {}
    for tag in ("one", "two"):
        sv = tmp_path / tag / "fern_style_style_nerf_relu_UseViewDir_ImgFactor8"
        sv.mkdir(parents=True)
        torch.save({"global_step": 7, "model": t(synth.nerf_state(0)), "model_fine": t(synth.nerf_state(1))}, sv / "000007.tar")
        torch.save({"global_step": 9, "model": t(synth.style_state(3)), "concat_model": t(synth.concat_state(2))}, sv / "style_000009.tar")
        outs[tag] = str(sv / "render_train_9")
    argv = ["--config", os.path.join(ROOT, "configs", "fern.txt"), "--datadir", str(scene), "--factor", "8",
            "--vae_pth_path", str(tmp_path / "vae.pth"), "--chunk", "1024", "--batch_size", "100", "--render_train_style"]
    state = torch.get_rng_state()
    torch.manual_seed(1234)
    try:
        assert train_tgtcs.main(argv + ["--basedir", str(tmp_path / "one")]) == outs["one"]
    finally:
        torch.set_rng_state(state)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TGTC_DIST_BACKEND")}
    try:
        mp.spawn(_cli_rank_seeded, args=(2, port, argv + ["--basedir", str(tmp_path / "two"), "--shard", shard], "gloo"), nprocs=2, join=True)
    finally:
        for k, v in env.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    names = sorted(os.listdir(outs["one"]))
    assert len(names) == 2 * 20 and sorted(os.listdir(outs["two"])) == names
    for n in names:
        assert open(os.path.join(outs["one"], n), "rb").read() == open(os.path.join(outs["two"], n), "rb").read(), n


def _batch_rank(rank, world, port, argv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), TGTC_DIST_BACKEND="gloo")
    from tgtc_style_amd import render_batch
    render_batch.main(argv)


def test_batch_render_of_scene_configs(tmp_path):
    """render_batch (BASELINE config 5): several scene configs in one job, two ranks with frames dealt round-robin; every
    scene's files are the ones the one-scene, one-rank CLI writes.  All five LLFF configs of the reference are rendered: four
    here, orchids by the single-process call at the end."""
    import torch.multiprocessing as mp
    from tgtc_style_amd import render_batch, train_tgtcs
    scenes = ["fern", "trex", "horns", "flower"]
    rest = ["--synthetic", "--synthetic_hw", "24", "--synthetic_frames", "3", "--chunk", "1024", "--batch_size", "576",
            "--render_valid_style", "--precision", "fp16"]          # config 5 names the fp16 path
    one = tmp_path / "one"
    ref = {s: train_tgtcs.main(["--config", os.path.join(ROOT, "configs", s + ".txt"), "--basedir", str(one)] + rest) for s in scenes}
    two = tmp_path / "two"
    argv = ["--configs"] + [os.path.join(ROOT, "configs", s + ".txt") for s in scenes] + ["--", "--basedir", str(two)] + rest
    mp.start_processes(_batch_rank, args=(2, 29531, argv), nprocs=2, join=True, start_method="spawn")
    for s in scenes:
        out = ref[s].replace(str(one), str(two))
        names = sorted(os.listdir(ref[s]))
        assert len(names) == 6 and sorted(os.listdir(out)) == names
        for n in names:
            assert open(os.path.join(out, n), "rb").read() == open(os.path.join(ref[s], n), "rb").read(), (s, n)
    # one process, no torchrun: the same loop
    outs = render_batch.main(["--configs", os.path.join(ROOT, "configs", "orchids.txt"), "--", "--basedir", str(tmp_path / "three")] + rest)
    assert len(outs) == 1 and len(os.listdir(outs[0])) == 6


def test_cli_whole_image_batches_equal_literal_batches(tmp_path):
    """The CLI renders whole images per call; --literal_batches feeds --batch_size rays at a time like the reference's
    loader.  Same files either way (device-generated rays, per-ray seeded jitter)."""
    from tgtc_style_amd import train_tgtcs
    base = ["--config", os.path.join(ROOT, "configs", "fern.txt"), "--synthetic", "--synthetic_hw", "40", "--synthetic_frames", "2",
            "--chunk", "1024", "--batch_size", "300", "--render_valid_style"]
    a = train_tgtcs.main(base + ["--basedir", str(tmp_path / "a")])
    b = train_tgtcs.main(base + ["--basedir", str(tmp_path / "b"), "--literal_batches"])
    names = sorted(os.listdir(a))
    assert len(names) == 4 and sorted(os.listdir(b)) == names
    for n in names:
        assert open(os.path.join(a, n), "rb").read() == open(os.path.join(b, n), "rb").read(), n


def test_cli_geometry_pass_two_ranks(tmp_path):
    """--render_valid (cal_geometry) under frames sharding: rank r renders and writes frames k = r (mod 2) with their global
    numbers, rank 0 assembles the scene-wide geometry.npz; every file equals the one-rank run's."""
    import socket
    import torch.multiprocessing as mp
    from tgtc_style_amd import train_tgtcs
    common = ["--config", os.path.join(ROOT, "configs", "fern.txt"), "--synthetic", "--synthetic_hw", "20", "--synthetic_frames", "3",
              "--chunk", "1024", "--batch_size", "128", "--render_valid"]
    one = train_tgtcs.main(common + ["--basedir", str(tmp_path / "one")])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TGTC_DIST_BACKEND")}
    try:
        mp.spawn(_cli_rank, args=(2, port, common + ["--basedir", str(tmp_path / "two")], "gloo"), nprocs=2, join=True)
    finally:
        for k, v in env.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    two = os.path.join(str(tmp_path / "two"), os.path.relpath(one, str(tmp_path / "one")))
    names = sorted(os.listdir(one))
    assert names == sorted(["rgb_%05d.png" % i for i in range(3)] + ["depth_%05d.png" % i for i in range(3)] +
                           ["geometry_%05d.npz" % i for i in range(3)] + ["geometry.npz"])
    assert sorted(os.listdir(two)) == names
    for n in names:
        if n.endswith(".png"):
            assert open(os.path.join(one, n), "rb").read() == open(os.path.join(two, n), "rb").read(), n
        else:
            a, b = np.load(os.path.join(one, n)), np.load(os.path.join(two, n))
            assert sorted(a.files) == sorted(b.files)
            for k in a.files:
                assert np.array_equal(a[k], b[k]), (n, k)


def test_cal_geometry_files_match_reference_written_files(tmp_path):
    """tests/golden/files/geometry_00001.npz and geometry.npz were written by the REFERENCE's cal_geometry
    (rendering.py:73,81; gen_golden.py g13_files) for 3 frames of 6x8 rays of the seeded nets.  This build's cal_geometry
    on the same inputs writes files with the same names, keys, dtypes and shapes, and the same numbers to 1e-3."""
    from tgtc_style_amd import config as cfg, models, rendering, synth, train_tgtcs, utils
    files = os.path.join(ROOT, "tests", "golden", "files")
    args = cfg.parse_args(["--config", os.path.join(ROOT, "configs", "fern.txt"), "--N_samples", "64", "--N_samples_fine", "64"])
    t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    nets = []
    for seed, mode in ((0, "coarse"), (1, "fine")):
        m = models.StyleNerf(args, mode=mode)
        m.load_state_dict(t(synth.nerf_state(seed)))
        nets.append(m.cuda())
    h, w, frames = 6, 8, 3
    rng = np.random.default_rng(1010)                                       # gen_golden.py test_rays(n, 1010)
    n = frames * h * w
    ro = np.concatenate([rng.uniform(-1.0, 1.0, (n, 2)), -np.ones((n, 1))], 1).astype(np.float64)
    rd = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), 2.0 * np.ones((n, 1))], 1).astype(np.float64)
    cps = np.tile(np.eye(4, dtype=np.float32)[None], (frames, 1, 1))
    cps[:, 0, 3] = np.arange(frames)

    class DS:
        mode, near, far = "train", 0., 1.
        hwf = [h, w, synth.fern_intrinsics(h, w)]
        frame_num = frames
    DS.h, DS.w, DS.cps, DS.cps_valid = h, w, cps, cps

    class Loader(list):
        dataset = DS
    batches = Loader({"rays_o": torch.from_numpy(ro[i:i + 40]), "rays_d": torch.from_numpy(rd[i:i + 40])} for i in range(0, n, 40))
    rgb_map, t_map = rendering.cal_geometry(dataloader=batches, sv_path=str(tmp_path), samp_func=utils.sampling_pts_uniform,
                                            samp_func_fine=utils.sampling_pts_fine_torch, args=args, device="cuda",
                                            model_forward=utils.batchify(lambda **k: nets[0](**k), 32),
                                            model_forward_fine=utils.batchify(lambda **k: nets[1](**k), 32),
                                            renderer=rendering.RayRenderer(*nets))
    # the rendered rays under the frozen parity criterion (tests/conditioning.py), the float32 oracle standing for the reference
    import conditioning
    from oracle import fields
    sds = [synth.nerf_state(0), synth.nerf_state(1)]
    td = lambda sd, dt: {k: v.to(dt) for k, v in t(sd).items()}
    tro, trd = torch.from_numpy(ro), torch.from_numpy(rd)
    render = lambda o, d, dc, df, sel, **kw: fields.render_plain(td(sds[0], dc), td(sds[1], df), o, d, 64, 64, dtype=dc, dtype_fine=df, **kw)
    e, ill = conditioning.check("cal_geometry 64c+64f", torch.as_tensor(rgb_map).reshape(-1, 3), torch.as_tensor(t_map).reshape(-1), render,
                              tro, trd, n_fine=64, max_certified=0.03,
                              stages=lambda sel: conditioning.hip_stages(nets[0], tro[sel].cuda(), trd[sel].cuda(), 64, 64))
    t32 = render(tro, trd, torch.float32, torch.float32, None)["t_fine"].numpy()
    strict = np.abs(np.asarray(torch.as_tensor(t_map).reshape(-1).cpu()) - t32) <= 1e-3        # rays on the float32 reference's own branch
    for name in ("geometry_00001.npz", "geometry.npz"):
        ref, mine = np.load(os.path.join(files, name)), np.load(os.path.join(str(tmp_path), name))
        assert sorted(mine.files) == sorted(ref.files), name
        for k in ref.files:
            assert mine[k].shape == ref[k].shape and mine[k].dtype == ref[k].dtype, (name, k, mine[k].dtype, ref[k].dtype)
            d = np.abs(mine[k].astype(np.float64) - ref[k].astype(np.float64))
            if k != "coor_map":
                assert d.max() <= 1e-6, (name, k, float(d.max()))
                continue
            # coor_map = o + t * d per pixel (|d| <= 2.1).  The file the REFERENCE wrote is the float32 oracle's output -- to 5e-8
            # where it was written (the build container: tests/test_oracle_golden.py::test_g13_geometry_file), but a float32
            # evaluation on ANOTHER CPU (this oracle runs on the GPU box's host) differs from it by up to 9e-4 here even on rays
            # whose variants agree on one machine, so across machines the file is held to the north-star tolerance.  On every ray
            # whose depth is within 1e-3 of the oracle the coordinates agree to 1e-3 * |d|; the other rays (the sampler's other
            # branch) are the ones conditioning.check has just accounted for, one by one
            rays = slice(48, 96) if name == "geometry_00001.npz" else slice(0, n)
            ref_pts = (ro[rays] + t32[rays, None] * rd[rays]).reshape(ref[k].shape)
            well = ~ill.numpy()[rays]
            dref = np.abs(ref[k] - ref_pts).reshape(-1, 3).max(1)
            print("%s: reference-written file vs the float32 oracle run HERE: %.2e on the %d well-conditioned rays, %.2e on the other %d" % (
                name, dref[well].max(), int(well.sum()), dref[~well].max() if (~well).any() else 0.0, int((~well).sum())))
            assert dref[well].max() <= 2.1e-3 and np.mean(dref <= 1e-5) >= 0.9, (name, "the reference-written file is not the float32 oracle's output", float(dref[well].max()))
            dr = d.reshape(-1, 3).max(1)
            assert dr[strict[rays]].max() <= 2.1e-3, (name, k, float(dr[strict[rays]].max()))
            assert strict[rays].mean() >= 0.95, (name, float(strict[rays].mean()))


@pytest.mark.parametrize("scene", ["fern", "flower", "horns", "orchids", "trex"])
def test_every_llff_config_inside_1e3_at_the_headline_precision(scene):
    """BASELINE config 5 names "all five LLFF configs, fp16 MFMA with fp32 alpha accumulation": each config file's own
    sample counts and network shape, a 40x40 frame of the scene's camera, rendered with fp16 MFMA operands in the fastest
    mode that holds the north-star tolerance (coarse fp16x3 + fine fp16mx, alpha compositing accumulated in fp32 / f64)
    and compared with the fp32 oracle at 1e-3 -- the plain `fp16` mode the batch test above renders is 3.5e-3 away."""
    from oracle import fields
    from tgtc_style_amd import config as cfg, models, rendering, synth, utils
    args = cfg.parse_args(["--config", os.path.join(ROOT, "configs", scene + ".txt"), "--precision", "fp16x3+fp16mx"])
    t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    seed = {"fern": 0, "flower": 30, "horns": 32, "orchids": 34, "trex": 36}[scene]
    sds = [synth.nerf_state(seed), synth.nerf_state(seed + 1)]
    nets = []
    for sd, mode in zip(sds, ("coarse", "fine")):
        m = models.StyleNerf(args, mode=mode)
        m.load_state_dict(t(sd))
        nets.append(m.cuda())
    assert nets[0].packed().precision == "fp16x3" and nets[1].packed().precision == "fp16mx"
    H = W = 40
    ro, rd = utils.gen_rays(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(seed + 3))
    out = rendering.RayRenderer(*nets).render(ro, rd, args.N_samples, args.N_samples_fine)
    import conditioning
    td = lambda sd, dt: {k: v.to(dt) for k, v in t(sd).items()}
    conditioning.check("%s %dc+%df" % (scene, args.N_samples, args.N_samples_fine), out["rgb"], out["t"],
                       lambda o, d, dc, df, sel, **kw: fields.render_plain(td(sds[0], dc), td(sds[1], df), o, d, args.N_samples, args.N_samples_fine,
                                                                           dtype=dc, dtype_fine=df, **kw), ro.cpu(), rd.cpu(), tol=1e-3,
                       n_fine=args.N_samples_fine,
                       stages=lambda sel: conditioning.hip_stages(nets[0], ro[sel.cuda()].contiguous(), rd[sel.cuda()].contiguous(),
                                                                  args.N_samples, args.N_samples_fine))
