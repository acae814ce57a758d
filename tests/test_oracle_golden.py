"""Pins the CPU oracle against golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only.

Tolerances: the oracle performs the same ATen ops in the same order as the reference, so most
checks are bit-exact (`equal`); the few that regroup arithmetic allow 1e-6.
"""
import numpy as np
import torch

from oracle import fields, raymarch, rays, style2d
from tgtc_style_amd import synth


def T(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def tt(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=0.0):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if tol == 0.0:
        assert torch.equal(a, b), float((a.double() - b.double()).abs().max())
    else:
        err = float((a.double() - b.double()).abs().max())
        assert err <= tol, err


def test_g1_rays(golden):
    g = golden("g1_rays")
    for tag, (H, W) in {"a": (12, 16), "b": (16, 16)}.items():
        focal = synth.fern_intrinsics(H, W)
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        for pi in (0, 37):
            for pa in (0, 1):
                key = "%s_p%d_%d" % (tag, pi, pa)
                o, d = rays.pinhole_rays(H, W, K, synth.spiral_pose(pi), bool(pa))
                o64, d64 = np.zeros([H, W, 3]), np.zeros([H, W, 3])
                o64[:], d64[:] = o, d
                close(o64, g[key + "_o"])
                close(d64, g[key + "_d"])
                no, nd = rays.ndc_warp(H, W, focal, 1., o64, d64)
                close(no, g[key + "_ndc_o"])
                close(nd, g[key + "_ndc_d"])
                if not pa:
                    fo, fd = rays.frame_rays_ndc(H, W, focal, synth.spiral_pose(pi))
                    close(fo, g[key + "_ndc_o"].reshape(-1, 3))
                    close(fd, g[key + "_ndc_d"].reshape(-1, 3))
                # after the warp every origin sits on the near plane
                assert np.allclose(no[..., 2], -1.0)


def test_g2_coarse(golden):
    g = golden("g2_coarse")
    ro, rd = tt(g["rays_o"]), tt(g["rays_d"])
    for n in (64, 128):
        pts, ts = raymarch.sample_coarse(ro, rd, n, 0., 1.)
        close(pts, g["pts_%d" % n])
        close(ts, g["ts_%d" % n])
        assert pts.dtype == torch.float64 and ts.dtype == torch.float32
        pts, ts = raymarch.sample_coarse(ro, rd, n, 0., 1., jitter=tt(g["jit_%d" % n]))
        close(pts, g["pts_jit_%d" % n])
        close(ts, g["ts_jit_%d" % n])


def test_g3_embed(golden):
    g = golden("g3_embed")
    x = tt(g["x"])
    close(fields.posenc(x, 10), g["pe10_f64"])
    close(fields.posenc(x, 4), g["pe4_f64"])
    close(fields.posenc(x.float(), 10), g["pe10_f32in"])
    assert fields.posenc(x, 10).shape[-1] == synth.PE_COOR and fields.posenc(x, 4).shape[-1] == synth.PE_DIR


def test_g4_nerf(golden):
    g = golden("g4_nerf")
    ro, rd, ts = tt(g["rays_o"]), tt(g["rays_d"]), tt(g["ts"])
    pts = ro[:, None, :] + ts[..., None] * rd[:, None, :]
    dirs = rd[:, None, :].expand(-1, ts.shape[1], -1)
    for name, seed in (("coarse", 0), ("fine", 1)):
        sd = T(synth.nerf_state(seed))
        out = fields.style_nerf(sd, pts, dirs)
        close(out["rgb"], g[name + "_rgb"])
        close(out["sigma"], g[name + "_sigma"])
        close(out["base_remap"][:, :48], g[name + "_remap_first48"])
        close(out["pts"][:, :8], g[name + "_pts_enc_first8"])
        close(out["dirs"][:, :8], g[name + "_dirs_enc_first8"])
        out2 = fields.nerf_mlp(sd, out["pts"], out["dirs"])
        close(out2["rgb"], g[name + "_mlp_rgb"])
        close(out2["sigma"], g[name + "_mlp_sigma"])


def test_g5_composite(golden):
    g = golden("g5_composite")
    for s in ("", "2"):
        r, t, w = raymarch.composite(tt(g["rgb" + s]), tt(g["sigma" + s]), tt(g["ts" + s]))
        close(r, g["rgb_exp" + s])
        close(t, g["t_exp" + s])
        close(w, g["weights" + s])
    # edge cases named in SURVEY G5
    w = g["weights"]
    assert np.all(w[1] == 0) and np.all(w[2] == 0)            # negative / zero density -> nothing
    assert abs(w[5, 0] - (1 - np.exp(-500.0 * (g["ts"][5, 1] - g["ts"][5, 0])))) < 1e-6


def test_g6_fine(golden):
    g = golden("g6_fine")
    for N in (64, 128):
        tag = "_%d" % N
        ro, rd, ts, w = (tt(g[k + tag]) for k in ("rays_o", "rays_d", "ts", "w"))
        mid = 0.5 * (ts[..., 1:] + ts[..., :-1])
        close(raymarch.inverse_cdf(mid, w[..., 1:-1], 64), g["samples" + tag])
        pts, tv = raymarch.sample_fine(ro, rd, ts, w, 64)
        close(tv, g["tvals" + tag])
        close(pts, g["pts" + tag])
        assert bool((tv[:, 1:] >= tv[:, :-1]).all())


def test_g7_style(golden):
    g = golden("g7_style")
    lat = T(synth.latents_state(4, style_num=2, frame_num=20))
    sid, fid = tt(g["style_ids"]), tt(g["frame_ids"])
    for sc in (0.0, 1.0, 0.35):
        close(fields.latents_forward(lat, sid, fid, sc), g["latents_s%g" % sc])
    x, z, conc = tt(g["x"]), tt(g["z"]), tt(g["conc"])
    close(fields.concat_mlp(T(synth.concat_state(2)), x, z)["concat_features"], g["concat_features"])
    close(fields.style_mlp(T(synth.style_state(3)), x, conc, z)["rgb"], g["style_rgb"])


def test_g8_end_to_end(golden):
    g = golden("g8_end_to_end")
    c, f = T(synth.nerf_state(0)), T(synth.nerf_state(1))
    cm, sm = T(synth.concat_state(2)), T(synth.style_state(3))
    lat = T(synth.latents_state(4))
    for nc, nf in ((128, 64), (64, 64)):
        tag = "_%dc%df" % (nc, nf)
        ro, rd = tt(g["rays_o" + tag]), tt(g["rays_d" + tag])
        # the reference ran through batchify(chunk=32) and 32-ray loader batches; rays are independent
        out = fields.render_plain(c, f, ro, rd, nc, nf)
        close(out["rgb_fine"], g["plain_rgb" + tag], 2e-6)
        close(out["t_fine"], g["plain_t" + tag], 2e-6)
        R = ro.shape[0]
        z = fields.latents_forward(lat, torch.zeros(R, dtype=torch.long), torch.full((R,), 33, dtype=torch.long), 1.0)
        for jt, jit in (("", None), ("_jit", tt(g["jit" + tag]))):
            out = fields.render_styled(c, f, cm, sm, ro, rd, z, nc, nf, jitter=jit)
            close(out["rgb_coarse"], g["styled_rgb_coarse" + jt + tag], 2e-6)
            close(out["ts_fine"], g["styled_ts_fine" + jt + tag], 1e-6)
            close(out["rgb_fine"], g["styled_rgb" + jt + tag], 2e-6)
            close(out["t_fine"], g["styled_t" + jt + tag], 2e-6)


def test_g9_style2d(golden):
    g = golden("g9_style2d")
    tsd = T(synth.transformer_state(5))
    close(style2d.mha(tsd, "decoder.layers.0.multihead_attn.", tt(g["mha_q"]), tt(g["mha_k"]), tt(g["mha_v"])),
          g["mha_out"], 2e-6)
    src, mem = tt(g["layer_src"]), tt(g["layer_mem"])
    close(style2d.encoder_layer(tsd, "encoder_s.layers.1.", src, has_pos=False), g["enc_s_out"], 5e-6)
    close(style2d.encoder_layer(tsd, "encoder_c.layers.2.", src, has_pos=True), g["enc_c_out"], 5e-6)
    close(style2d.decoder_layer(tsd, "decoder.layers.1.", src, mem, src * 0.5), g["declayer_out"], 5e-6)
    close(style2d.transformer_forward(tsd, tt(g["tr_style"]), tt(g["tr_content"])), g["tr_hs"], 2e-5)

    img = tt(g["img"])
    esd, dsd, vsd = T(synth.embed_state(6)), T(synth.decoder_state(7)), T(synth.vgg_state(8))
    close(style2d.patch_embed(esd, img), g["embed_out"], 1e-6)
    close(style2d.cnn_decode(dsd, tt(g["cnn_in"])), g["cnn_out"], 1e-5)
    feats = style2d.vgg_encode(vsd, img)
    for i in range(4):
        close(feats[i], g["vgg_%d" % (i + 1)], 1e-5)
    assert torch.equal(feats[3], feats[4])
    m, s = style2d.mean_std(feats[2])
    close(m, g["ms_mean"], 1e-6)
    close(s, g["ms_std"], 1e-6)
    close(style2d.adain(feats[1], feats[1].flip(-1) * 0.5 + 0.1), g["adain"], 1e-5)

    out = style2d.stylize(esd, tsd, dsd, tt(g["st_content"]), tt(synth.style_image(11, 40, 56)))
    close(out["hs"], g["st_hs"], 2e-5)
    close(out["ics"], g["st_ics"], 2e-4)
    close(out["image"], g["st_image"], 2e-4)
    close(out["style_feature"], g["st_feature"], 2e-5)


def test_g10_image_epilogue(golden):
    """oracle.image vs the arrays the reference's cal_geometry handed to imageio.imwrite (integer work: bit exact)."""
    from oracle import image
    g = golden("g10_image")
    frames, h, w = int(g["frames"]), int(g["h"]), int(g["w"])
    rgb8, depth8 = image.frames_to_uint8(g["rgb"], g["t"], frames)
    assert rgb8.dtype == np.uint8 and depth8.dtype == np.uint8
    assert np.array_equal(rgb8.reshape(frames, h, w, 3), g["rgb8"])
    assert np.array_equal(depth8.reshape(frames, h, w), g["depth8"])


def test_g12_vae_encode(golden):
    """oracle.fields.vae_encode vs the reference's VAE.encode(x, various=False) (same ATen ops: bit exact); the VAE
    container of the package carries the reference's state-dict keys."""
    from tgtc_style_amd import models
    g = golden("g12_vae")
    mu, logvar = fields.vae_encode(T(synth.vae_state(9)), tt(g["style_features"]))
    close(mu, g["mu"])
    close(logvar, g["logvar"])
    vae = models.VAE(data_dim=1024, latent_dim=32, W=512, D=4)
    assert sorted(vae.state_dict().keys()) == list(g["state_keys"])
    vae.load_state_dict(T(synth.vae_state(9)))          # strict


def _oracle_train_step(g, dtype, golden_depths=False):
    """The oracle's formulas for the Origin_train body (train_tgtcs.py:226-254) with autograd, on g14's stored inputs.
    golden_depths: evaluate the fine half at the fine depths the reference sampled (the sampler is not differentiated, and a
    chain in another precision may take another branch of it on an ill-conditioned ray)."""
    R, N, NF = g["rays_o"].shape[0], int(g["n_coarse"]), int(g["n_fine"])
    ro, rd, gt = tt(g["rays_o"]), tt(g["rays_d"]), tt(g["rgb_gt"]).to(dtype)
    w = [{k: v.clone().to(dtype).requires_grad_() for k, v in T(synth.nerf_state(int(s))).items()} for s in g["seeds"]]

    def net(sd, pts, n):
        pe = fields.posenc(pts, 10).to(dtype).reshape(R * n, -1)
        de = fields.posenc(rd[:, None, :].expand(R, n, 3), 4).to(dtype).reshape(R * n, -1)
        ret = fields.nerf_mlp(sd, pe, de)
        return ret["rgb"].reshape(R, n, 3), ret["sigma"].reshape(R, n)
    pts, ts = raymarch.sample_coarse(ro, rd, N, 0., 1., jitter=tt(g["jitter"]))
    rgb, sig = net(w[0], pts, N)
    rgb_c, _, w_c = raymarch.composite(rgb, sig + tt(g["noise_coarse"]).to(dtype), ts.to(dtype))
    pts_f, ts_f = raymarch.sample_fine(ro, rd, ts, w_c.detach().float(), NF)      # the sampler carries no gradient (utils.py:562-579)
    if golden_depths:
        ts_f = tt(g["ts_fine"])
        pts_f = ro[:, None, :] + rd[:, None, :] * ts_f[..., None]
    rgb, sig = net(w[1], pts_f, N + NF)
    rgb_f, _, _ = raymarch.composite(rgb, sig + tt(g["noise_fine"]).to(dtype), ts_f.to(dtype))
    l_c, l_f = ((rgb_c - gt) ** 2).mean(), ((rgb_f - gt) ** 2).mean()
    (l_c + l_f).backward()
    return dict(ts=ts, ts_fine=ts_f, rgb_exp=rgb_c, rgb_exp_fine=rgb_f, loss_rgb=l_c, loss_rgb_fine=l_f,
                grads=[{k: v.grad for k, v in sd.items()} for sd in w])


def test_g14_origin_train_step(golden):
    """g14 = one Origin_train iteration run by the reference itself (gen_golden.py g14_train: its samplers, networks,
    compositing, losses and torch autograd).  The oracle with autograd reproduces the losses and all 48 gradients: this pins
    the yardstick the GPU gradient tests use (tests/test_hip_backward_gpu.py)."""
    g = golden("g14_train")
    o = _oracle_train_step(g, torch.float32)
    close(o["ts"], g["ts"], 1e-7)
    close(o["ts_fine"], g["ts_fine"], 2e-6)
    close(o["rgb_exp"].detach(), g["rgb_exp"], 2e-6)
    close(o["rgb_exp_fine"].detach(), g["rgb_exp_fine"], 2e-6)
    for k in ("loss_rgb", "loss_rgb_fine"):
        assert abs(float(o[k]) - float(g[k])) <= 1e-6 * float(g[k]), k
    worst = 0.0
    for tag, grads in zip(("coarse", "fine"), o["grads"]):
        for k, v in grads.items():
            ref = tt(g["grad_%s.%s" % (tag, k)])
            err = float((v - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
            worst = max(worst, err)
            assert err <= 2e-4, (tag, k, err)      # two float32 autograd runs of the same formulas (summation order)
    print("oracle float32 autograd vs the reference's: worst relative gradient difference %.2e" % worst)


def test_g14_coherence_term(golden):
    """The coherence term of Style_train (train_tgtcs.py:394-403, :451-458; VGGNet.py:204-210, utils.py L2_norm) as the host
    code of this build computes it (training.CoherenceState), against the reference's value and autograd gradients."""
    from tgtc_style_amd import training
    g = golden("g14_train")
    st = training.CoherenceState(frame_num=20)
    st.cnt, st.x, st.y, st.x_origin = 1, tt(g["coh_x_prev"]), tt(g["coh_y_prev"]), tt(g["coh_x_origin_prev"])
    rgb2, rgb_fine2 = tt(g["coh_rgb2"]).clone().requires_grad_(), tt(g["coh_rgb_fine2"]).clone().requires_grad_()
    origin2 = tt(g["coh_rgb_origin2"])
    loss = st.coarse(rgb2, origin2) + st.fine(rgb_fine2, origin2)
    loss.backward()
    assert abs(float(loss) - float(g["coh_loss"])) <= 1e-6 * abs(float(g["coh_loss"]))
    close(rgb2.grad, g["coh_grad_rgb2"], 1e-6)
    close(rgb_fine2.grad, g["coh_grad_rgb_fine2"], 1e-6)
    assert st.cnt == 2 and torch.equal(st.x, rgb2.detach()) and torch.equal(st.y, rgb_fine2.detach())


def test_g13_geometry_file():
    """tests/golden/files/geometry*.npz were written by the REFERENCE's cal_geometry (gen_golden.py g13_files).  The oracle
    reproduces them: coor_map = o + t * d with the float32 oracle's depth -- to 5e-8 on the CPU that wrote them; another CPU's
    float32 evaluation may take another branch of the sampler on a few rays (tests/conditioning.py), hence a share."""
    import os
    files = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "files")
    h, w, frames = 6, 8, 3
    rng = np.random.default_rng(1010)
    n = frames * h * w
    ro = np.concatenate([rng.uniform(-1.0, 1.0, (n, 2)), -np.ones((n, 1))], 1).astype(np.float64)
    rd = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), 2.0 * np.ones((n, 1))], 1).astype(np.float64)
    t32 = fields.render_plain(T(synth.nerf_state(0)), T(synth.nerf_state(1)), tt(ro), tt(rd), 64, 64)["t_fine"].numpy()
    for name, rays in (("geometry_00001.npz", slice(48, 96)), ("geometry.npz", slice(0, n))):
        ref = np.load(os.path.join(files, name))["coor_map"]
        d = np.abs(ref - (ro[rays] + t32[rays, None] * rd[rays]).reshape(ref.shape)).reshape(-1, 3).max(1)
        print(name, "max %.2e, %d of %d rays within 1e-5" % (d.max(), int((d <= 1e-5).sum()), d.size))
        assert np.mean(d <= 1e-5) >= 0.9 and d.max() <= 0.1
