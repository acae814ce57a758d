"""CPU: host-side file formats either side of the hot path -- the image side of the
LLFF loader and the stylized_data.npz / image-transform plumbing of the 2-D pass driver.  No GPU, no compute kernels."""
import os

import numpy as np
import pytest

from tgtc_style_amd import llff_images, trans_test

PIL = pytest.importorskip("PIL.Image")


def _scene(tmp_path, n=3, h=24, w=36, with_minified=None):
    rng = np.random.default_rng(0)
    base = tmp_path / "scene"
    (base / "images").mkdir(parents=True)
    imgs = []
    for i in range(n):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        PIL.fromarray(a).save(base / "images" / ("img_%02d.png" % i))
        imgs.append(a)
    poses = rng.standard_normal((n, 3, 5))
    poses[:, 0, 4], poses[:, 1, 4], poses[:, 2, 4] = h, w, 30.0
    np.save(base / "poses_bounds.npy", np.concatenate([poses.reshape(n, 15), np.tile([[1.0, 9.0]], (n, 1))], 1))
    if with_minified:
        d = base / ("images_%s" % with_minified)
        d.mkdir()
        small = [rng.integers(0, 256, (h // 2, w // 2, 4), dtype=np.uint8) for _ in range(n)]     # RGBA: alpha is dropped
        for i, a in enumerate(small):
            PIL.fromarray(a).save(d / ("img_%02d.png" % i))
        return str(base), imgs, small
    return str(base), imgs, None


def test_load_data_uses_an_existing_minified_directory_exactly(tmp_path):
    base, _, small = _scene(tmp_path, with_minified=2)
    poses, bds, imgs = llff_images.load_data(base, factor=2)
    assert poses.shape == (3, 5, 3) and bds.shape == (2, 3) and imgs.shape == (12, 18, 3, 3)
    # load_llff.py:95-97: (H, W) of the table = the minified image, focal / factor
    assert np.all(poses[0, 4] == 12) and np.all(poses[1, 4] == 18) and np.allclose(poses[2, 4], 15.0)
    for i, a in enumerate(small):
        assert np.array_equal(imgs[..., i], a[..., :3] / 255.)           # load_llff.py:108: imread(f)[..., :3] / 255.
    images = llff_images.load_images(base, 2)
    assert images.shape == (3, 12, 18, 3) and images.dtype == np.float32


def test_minify_creates_the_cache_once(tmp_path):
    base, full, _ = _scene(tmp_path)
    d = llff_images.minify(base, 2)
    files = sorted(os.listdir(d))
    assert d.endswith("images_2") and files == ["img_00.png", "img_01.png", "img_02.png"]
    assert PIL.open(os.path.join(d, files[0])).size == (18, 12)
    stamp = os.path.getmtime(os.path.join(d, files[0]))
    assert llff_images.minify(base, 2) == d and os.path.getmtime(os.path.join(d, files[0])) == stamp
    poses, bds = llff_images.load_data(base, factor=2, load_imgs=False)
    assert np.all(poses[0, 4] == 12) and np.allclose(poses[2, 4], 15.0)
    # factor None: the full-size images, table untouched apart from (H, W)
    poses, bds, imgs = llff_images.load_data(base, factor=None)
    assert imgs.shape == (24, 36, 3, 3) and np.array_equal(imgs[..., 1], full[1] / 255.) and np.allclose(poses[2, 4], 30.0)


def test_mismatch_between_images_and_poses_is_an_error(tmp_path):
    base, _, _ = _scene(tmp_path, with_minified=2)
    os.remove(os.path.join(base, "images_2", "img_02.png"))
    with pytest.raises(ValueError):
        llff_images.load_data(base, factor=2)


def test_stylized_data_round_trip(tmp_path):
    """The npz the 2-D driver writes (trans_test.py:179) read back the way the datasets do (dataset.py:437-440)."""
    out = tmp_path / "data" / "stylized_gen_4.0"
    out.mkdir(parents=True)
    feats = np.arange(1024, dtype=np.float32)[None]
    style = np.random.default_rng(1).random((1, 512, 512, 3)).astype(np.float32)
    np.savez(os.path.join(out, "stylized_data"), style_names={"starry": 0}, style_paths="style/starry.jpg",
             style_images=style, style_features=feats)
    d = trans_test.read_stylized_data(str(tmp_path / "data"), 4.0)
    assert d["style_names"] == {"starry": 0} and d["style_num"] == 1 and str(d["style_paths"]) == "style/starry.jpg"
    assert np.array_equal(d["style_features"], feats) and np.array_equal(d["style_images"], style)
    assert trans_test.read_stylized_data(str(tmp_path / "data"), 8.0) is None


def test_image_transforms_match_torchvision_semantics():
    a = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    img = PIL.fromarray(a)
    t = trans_test._to_tensor(img)
    assert t.shape == (3, 5, 7) and float(t[1, 2, 3]) == a[2, 3, 1] / np.float32(255.0)
    # CenterCrop inside the image: offsets int(round((H - h) / 2)), int(round((W - w) / 2))
    c = np.asarray(trans_test._center_crop(img, 3, 4))
    assert np.array_equal(c, a[1:4, 2:6])
    # crop window larger than the image: centred, black outside
    c = np.asarray(trans_test._center_crop(img, 7, 9))
    assert c.shape == (7, 9, 3) and np.array_equal(c[1:6, 1:8], a) and not c[0].any() and not c[:, 0].any()
    # an excess of 3 pixels: torchvision pads left / top (3 // 2) = 1 and right / bottom 2 (not the banker's-rounded -1.5 -> -2)
    c = np.asarray(trans_test._center_crop(img, 8, 10))
    assert c.shape == (8, 10, 3) and np.array_equal(c[1:6, 1:8], a) and not c[0].any() and not c[6:].any() and not c[:, 8:].any()
    s = trans_test.style_image_array.__doc__
    assert "512" in s


def test_image_writer_background_files(tmp_path):
    """image_writer.ImageWriter: files appear after drain(), pixels unchanged, a failing file surfaces at drain()."""
    import torch
    from PIL import Image
    from tgtc_style_amd import image_writer as iw
    w = iw.ImageWriter(workers=3)
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (37, 53, 3), dtype=np.uint8) for _ in range(6)] + [rng.integers(0, 256, (20, 31), dtype=np.uint8)]
    for i, a in enumerate(imgs):
        w.save(str(tmp_path / ("%03d.png" % i)), a if i % 2 else torch.from_numpy(a))
    w.drain()
    for i, a in enumerate(imgs):
        assert np.array_equal(np.asarray(Image.open(tmp_path / ("%03d.png" % i))), a)
    w.save(str(tmp_path / "no_such_dir" / "x.png"), imgs[0])
    w.save(str(tmp_path / "ok.png"), imgs[1])
    with pytest.raises(OSError):
        w.drain()
    assert (tmp_path / "ok.png").exists()       # the other files of the batch are still written
    w.drain()                                   # nothing pending: returns


def test_checkpoint_files_round_trip(tmp_path):
    """checkpoints.py: the reference's file names / dictionary layouts, its newest-file selection and its rotation."""
    import torch
    from tgtc_style_amd import checkpoints as ck, models, synth
    t = lambda sd: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}

    class A:
        use_viewdir, act_type, embed_freq_coor, embed_freq_dir = True, "relu", 10, 4
        netdepth = netdepth_fine = 8
        netwidth = netwidth_fine = 256
        style_D, vae_latent, precision = 8, 32, "fp16x3"
    m, mf = models.StyleNerf(A, mode="coarse"), models.StyleNerf(A, mode="fine")
    m.load_state_dict(t(synth.nerf_state(0))), mf.load_state_dict(t(synth.nerf_state(1)))
    cm, sm = models.StyleMLP_before_concat(A), models.StyleMLP_Wild_multilayers(A)
    cm.load_state_dict(t(synth.concat_state(2))), sm.load_state_dict(t(synth.style_state(3)))
    lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    d = str(tmp_path / "ckpts")
    assert ck.load_nerf(d, m, mf) is None and ck.load_style(d, sm, cm) is None and not ck.load_latents(d, lat)
    for step in (500, 1000, 1500):
        p = ck.save_nerf(d, step, m, mf, keep=2)
    assert os.path.basename(p) == "001500.tar" and sorted(os.listdir(d)) == ["001000.tar", "001500.tar"]       # rotated
    sd = torch.load(p)
    assert sorted(sd) == ["global_step", "model", "model_fine", "optimizer", "style_optimizer"]
    ck.save_style(d, 120500, sm, cm)
    ck.save_latents(d, 120500, lat)
    assert sorted(os.listdir(d)) == ["001000.tar", "001500.tar", "latent_120500.tar", "style_120500.tar"]
    m2, mf2 = models.StyleNerf(A, mode="coarse"), models.StyleNerf(A, mode="fine")
    assert ck.load_nerf(d, m2, mf2) == 1500                      # the newest NeRF file, not the style / latent ones
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    cm2, sm2 = models.StyleMLP_before_concat(A), models.StyleMLP_Wild_multilayers(A)
    assert ck.load_style(d, sm2, cm2) == 120500
    assert all(torch.equal(a, b) for a, b in zip(sm.state_dict().values(), sm2.state_dict().values()))
    lat2 = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    assert ck.load_latents(d, lat2) and torch.equal(lat.latents, lat2.latents)
    # the 2-D module's three files are what trans_test.load_network reads back
    from tgtc_style_amd import style2d
    tr, dec, emb = style2d.Transformer(), style2d.Decoder(), style2d.PatchEmbed()
    s2 = str(tmp_path / "pretrained")
    ck.save_style2d(s2, 160000, tr, dec, emb)
    assert sorted(os.listdir(s2)) == ["decoder_iter_160000.pth", "embedding_iter_160000.pth", "transformer_iter_160000.pth"]
    assert sorted(torch.load(os.path.join(s2, "decoder_iter_160000.pth"))) == ["decoder", "step"]
    assert trans_test._newest(s2, "transformer").endswith("transformer_iter_160000.pth")


FILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "files")


def test_reference_written_checkpoint_files(tmp_path):
    """tests/golden/files/*.tar hold the state dicts of the REFERENCE's own modules and optimisers under the keys of its
    save sites (gen_golden.py g13_files; train_tgtcs.py:284-300, :503-517): checkpoints.py finds, reads and loads them
    into this build's modules -- same key names, same shapes -- and a file this build writes has the same layout."""
    import shutil
    import torch
    from tgtc_style_amd import checkpoints as ck, models

    class A:
        use_viewdir, act_type, embed_freq_coor, embed_freq_dir = True, "relu", 10, 4
        netdepth = netdepth_fine = 8
        netwidth = netwidth_fine = 8
        style_D, vae_latent, precision = 8, 32, "fp16x3"

    for f in ("000500.tar", "style_120500.tar", "latent_120500.tar"):
        shutil.copy(os.path.join(FILES, f), tmp_path / f)
    model, fine = models.StyleNerf(A, mode="coarse"), models.StyleNerf(A, mode="fine")
    assert ck.load_nerf(str(tmp_path), model, fine) == 500                       # train_tgtcs.py:60-72
    ref = torch.load(os.path.join(FILES, "000500.tar"), map_location="cpu")
    assert sorted(ref) == ["global_step", "model", "model_fine", "optimizer", "style_optimizer"]
    for name, m in (("model", model), ("model_fine", fine)):
        assert list(m.state_dict()) == list(ref[name])
        for k, v in m.state_dict().items():
            assert torch.equal(v, ref[name][k]), (name, k)
    concat, style = models.StyleMLP_before_concat(A), models.StyleMLP_Wild_multilayers(A)
    assert ck.load_style(str(tmp_path), style, concat) == 120500                 # :74-82
    ref_s = torch.load(os.path.join(FILES, "style_120500.tar"), map_location="cpu")
    assert sorted(ref_s) == ["concat_model", "global_step", "model", "optimizer"]
    assert all(torch.equal(v, ref_s["model"][k]) for k, v in style.state_dict().items())
    assert all(torch.equal(v, ref_s["concat_model"][k]) for k, v in concat.state_dict().items())
    lat = models.StyleLatents_variational(style_num=2, frame_num=3, latent_dim=32)
    assert ck.load_latents(str(tmp_path), lat)                                   # :139-146
    ref_l = torch.load(os.path.join(FILES, "latent_120500.tar"), map_location="cpu")
    assert sorted(ref_l) == ["global_step", "train_set_1"] and list(lat.state_dict()) == list(ref_l["train_set_1"])
    assert torch.equal(lat.latents.detach(), ref_l["train_set_1"]["latents"])
    # what this build writes has the reference file's layout, key for key (optimiser state dicts included)
    opt = torch.optim.Adam(list(model.parameters()) + list(fine.parameters()), lr=5e-4, betas=(0.9, 0.999))
    sopt = torch.optim.Adam(list(style.parameters()) + list(concat.parameters()), lr=5e-4, betas=(0.9, 0.999))
    for o, mods in ((opt, (model, fine)), (sopt, (style, concat))):
        sum((p ** 2).sum() for m in mods for p in m.parameters()).backward()
        o.step()
    mine = torch.load(ck.save_nerf(str(tmp_path / "w"), 500, model, fine, opt, sopt), map_location="cpu")
    assert sorted(mine) == sorted(ref) and list(mine["model"]) == list(ref["model"])
    assert sorted(mine["optimizer"]) == sorted(ref["optimizer"]) and \
        [g["params"] for g in mine["optimizer"]["param_groups"]] == [g["params"] for g in ref["optimizer"]["param_groups"]]
    mine_s = torch.load(ck.save_style(str(tmp_path / "w"), 120500, style, concat, sopt), map_location="cpu")
    assert sorted(mine_s) == sorted(ref_s) and list(mine_s["concat_model"]) == list(ref_s["concat_model"])
    mine_l = torch.load(ck.save_latents(str(tmp_path / "w"), 120500, lat), map_location="cpu")
    assert sorted(mine_l) == sorted(ref_l) and list(mine_l["train_set_1"]) == list(ref_l["train_set_1"])


def test_reference_written_npz_files(tmp_path):
    """geometry_00001.npz / geometry.npz as the reference's cal_geometry wrote them (rendering.py:73,81) and a
    stylized_data.npz in trans_test.py:179's layout: names, dtypes and shapes this build's readers and writers rely on."""
    import shutil
    g1 = np.load(os.path.join(FILES, "geometry_00001.npz"))
    assert sorted(g1.files) == ["coor_map", "cps", "far", "hwf", "near"]
    assert g1["coor_map"].shape == (6, 8, 3) and g1["coor_map"].dtype == np.float32
    assert g1["cps"].shape == (4, 4) and g1["cps"].dtype == np.float32 and float(g1["cps"][0, 3]) == 1.0   # frame 1's pose
    assert g1["hwf"].shape == (3,) and float(g1["hwf"][0]) == 6 and float(g1["hwf"][1]) == 8
    assert float(g1["near"]) == 0.0 and float(g1["far"]) == 1.0
    g = np.load(os.path.join(FILES, "geometry.npz"))
    assert sorted(g.files) == sorted(g1.files) and g["coor_map"].shape == (3, 6, 8, 3) and g["cps"].shape == (3, 4, 4)
    assert np.array_equal(g["coor_map"][1], g1["coor_map"])
    dst = tmp_path / "data" / "stylized_gen_8.0"
    dst.mkdir(parents=True)
    shutil.copy(os.path.join(FILES, "stylized_data.npz"), dst / "stylized_data.npz")
    d = trans_test.read_stylized_data(str(tmp_path / "data"), 8.0)                 # dataset.py:437-440
    assert d["style_names"] == {"starry_night": 0} and str(d["style_paths"]) == "./style/starry_night.jpg"
    assert d["style_images"].shape == (1, 16, 16, 3) and d["style_features"].shape == (1, 1024) and d["style_num"] == 1
    assert d["style_features"].dtype == np.float32
