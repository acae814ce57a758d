#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE itself.

Run in the build container only (needs /root/reference):
    python tests/golden/gen_golden.py

The reference publishes no tests or golden vectors, so parity is pinned by
importing its modules here (with inert placeholders for the third-party packages that are not
installed, see ref_shim.py), feeding them seeded synthetic weights / rays / images, and
recording inputs + outputs as data.  No reference source is copied; the fixtures are arrays.

Inputs that come from a seeded generator in tgtc_style_amd.synth are stored as seeds/arguments,
not arrays, to keep the fixtures small; tests regenerate them bit-identically.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

_saved = ref_shim.install()
import dataset as ref_dataset  # noqa: E402
import function as ref_function  # noqa: E402
import models as ref_models  # noqa: E402
import rendering as ref_rendering  # noqa: E402
import Style_function as ref_style_function  # noqa: E402
import tctrans as ref_tctrans  # noqa: E402
import transformer as ref_transformer  # noqa: E402
import utils as ref_utils  # noqa: E402

ref_shim.restore_env(_saved)

from tgtc_style_amd import synth  # noqa: E402

torch.set_num_threads(8)


class Args:
    """The argparse fields the reference constructors read (config.py defaults + configs/fern.txt)."""
    use_viewdir = True
    act_type = "relu"
    embed_freq_coor = 10
    embed_freq_dir = 4
    netdepth = 8
    netwidth = 256
    netdepth_fine = 8
    netwidth_fine = 256
    style_D = 8
    vae_latent = 32
    siren_sigma_mul = 20.0
    N_samples = 64
    N_samples_fine = 64
    dataset_type = "llff"


def t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-22s %8.1f KB  %s" % (name, os.path.getsize(path) / 1024, sorted(out)))


def make_nerf(seed, mode, args=Args):
    m = ref_models.StyleNerf(args, mode=mode).eval()
    m.load_state_dict(t(synth.nerf_state(seed)))
    return m


def test_rays(n, seed, spread=1.0):
    """Seeded NDC-like rays: origins on the near plane (oz=-1), d_z = 2 (after the NDC warp)."""
    rng = np.random.default_rng(seed)
    o = np.concatenate([rng.uniform(-spread, spread, (n, 2)), -np.ones((n, 1))], 1)
    d = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), 2.0 * np.ones((n, 1))], 1)
    return o.astype(np.float64), d.astype(np.float64)


# ----------------------------------------------------------------------------- G1 rays
def g1_rays():
    out = {}
    for tag, (H, W) in {"a": (12, 16), "b": (16, 16)}.items():
        focal = synth.fern_intrinsics(H, W)
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        for pi in (0, 37):
            c2w = synth.spiral_pose(pi)
            for pa in (False, True):
                ro, rd = ref_dataset.get_rays_np(H, W, K, c2w, pa)
                # dataset.py:420-433 stores into float64 buffers, then warps
                ro64, rd64 = np.zeros([H, W, 3]), np.zeros([H, W, 3])
                ro64[:], rd64[:] = ro, rd
                no, nd = ref_dataset.ndc_rays_np(H, W, K[0][0], 1., ro64, rd64)
                key = "%s_p%d_%d" % (tag, pi, int(pa))
                out[key + "_o"], out[key + "_d"] = ro64, rd64
                out[key + "_ndc_o"], out[key + "_ndc_d"] = no, nd
    save("g1_rays", **out)


# ----------------------------------------------------------------------------- G2 coarse sampling
def g2_coarse():
    out = {}
    ro, rd = test_rays(16, 100)
    out["rays_o"], out["rays_d"] = ro, rd
    for n in (64, 128):
        pts, ts = ref_utils.sampling_pts_uniform(torch.from_numpy(ro), torch.from_numpy(rd), N_samples=n,
                                                 near=0., far=1., perturb=False)
        out["pts_%d" % n], out["ts_%d" % n] = pts, ts
        torch.manual_seed(1234 + n)
        pts, ts = ref_utils.sampling_pts_uniform(torch.from_numpy(ro), torch.from_numpy(rd), N_samples=n,
                                                 near=0., far=1., perturb=True)
        torch.manual_seed(1234 + n)   # the jitter is the function's first RNG draw (utils.py:519-520)
        jit = torch.zeros([16, n])
        torch.nn.init.uniform_(jit, 0, 1)
        out["jit_%d" % n], out["pts_jit_%d" % n], out["ts_jit_%d" % n] = jit, pts, ts
    save("g2_coarse", **out)


# ----------------------------------------------------------------------------- G3 embedder
def g3_embed():
    rng = np.random.default_rng(3)
    x = rng.uniform(-1.5, 1.5, (40, 3))
    x[0] = 0.0
    x[1] = 1.0
    x[2] = -1.0
    x[3] = [1.5, -1.5, 0.5]
    x = torch.from_numpy(x)
    e10 = ref_models.Embedder(input_dim=3, max_freq_log2=9, N_freqs=10)
    e4 = ref_models.Embedder(input_dim=3, max_freq_log2=3, N_freqs=4)
    save("g3_embed", x=x, pe10_f64=e10(x), pe4_f64=e4(x), pe10_f32in=e10(x.float()), pe4_f32in=e4(x.float()))


# ----------------------------------------------------------------------------- G4 NeRF MLP
def g4_nerf():
    out = {}
    ro, rd = test_rays(4, 400)
    ro_t, rd_t = torch.from_numpy(ro), torch.from_numpy(rd)
    rng = np.random.default_rng(401)
    ts = np.sort(rng.uniform(0, 1, (4, 192)).astype(np.float32), -1)
    pts = ro_t[:, None, :] + torch.from_numpy(ts)[..., None] * rd_t[:, None, :]
    dirs = rd_t[:, None, :].expand(-1, 192, -1)
    out["rays_o"], out["rays_d"], out["ts"] = ro, rd, ts
    for name, seed, mode in (("coarse", 0, "coarse"), ("fine", 1, "fine")):
        m = make_nerf(seed, mode)
        with torch.no_grad():
            ret = m(pts=pts, dirs=dirs)
        out[name + "_rgb"], out[name + "_sigma"] = ret["rgb"], ret["sigma"]
        out[name + "_remap_first48"] = ret["base_remap"][:, :48]
        out[name + "_pts_enc_first8"] = ret["pts"][:, :8]
        out[name + "_dirs_enc_first8"] = ret["dirs"][:, :8]
        # the granular MLP_style.forward on already-encoded float32 inputs (models.py:95)
        with torch.no_grad():
            ret2 = m.net(pts=ret["pts"], dirs=ret["dirs"])
        out[name + "_mlp_rgb"], out[name + "_mlp_sigma"] = ret2["rgb"], ret2["sigma"]
    save("g4_nerf", **out)


# ----------------------------------------------------------------------------- G5 compositing
def g5_composite():
    rng = np.random.default_rng(5)
    R, N = 12, 64
    rgb = rng.uniform(0, 1, (R, N, 3)).astype(np.float32)
    sigma = (rng.standard_normal((R, N)) * 20).astype(np.float32)
    ts = np.sort(rng.uniform(0, 1, (R, N)).astype(np.float32), -1)
    sigma[1] = -np.abs(sigma[1])          # all negative -> zero weights
    sigma[2] = 0.0                        # all zero
    sigma[3, 10] = 1e6                    # one huge density
    ts[4, 20:24] = ts[4, 20]              # repeated depths -> zero deltas
    sigma[5] = 500.0                      # opaque from the first sample
    ts[6] = np.linspace(0, 1, N, dtype=np.float32)
    out = {"rgb": rgb, "sigma": sigma, "ts": ts}
    r, te, w = ref_utils.alpha_composition(torch.from_numpy(rgb), torch.from_numpy(sigma), torch.from_numpy(ts), 0)
    out["rgb_exp"], out["t_exp"], out["weights"] = r, te, w
    # a 192-sample case (fine pass length)
    N2 = 192
    rgb2 = rng.uniform(0, 1, (6, N2, 3)).astype(np.float32)
    sigma2 = (rng.standard_normal((6, N2)) * 40).astype(np.float32)
    ts2 = np.sort(rng.uniform(0, 1, (6, N2)).astype(np.float32), -1)
    r, te, w = ref_utils.alpha_composition(torch.from_numpy(rgb2), torch.from_numpy(sigma2), torch.from_numpy(ts2), 0)
    out.update(rgb2=rgb2, sigma2=sigma2, ts2=ts2, rgb_exp2=r, t_exp2=te, weights2=w)
    save("g5_composite", **out)


# ----------------------------------------------------------------------------- G6 fine sampling
def g6_fine():
    rng = np.random.default_rng(6)
    out = {}
    for N, NF in ((64, 64), (128, 64)):
        R = 10
        ro, rd = test_rays(R, 600 + N)
        ts = torch.linspace(0, 1, N).expand(R, N).contiguous().numpy().copy()
        ts[8] = np.sort(rng.uniform(0, 1, N).astype(np.float32))      # jittered depths
        ts[9] = np.sort(rng.uniform(0, 1, N).astype(np.float32))
        w = rng.uniform(0, 1, (R, N)).astype(np.float32) ** 4
        w[1] = 0.0                                   # all zero -> uniform pdf
        w[2] = 0.0; w[2, 1] = 1.0                    # spike in the first interior bin
        w[3] = 0.0; w[3, N - 2] = 1.0                # spike in the last interior bin
        w[4] = 0.0; w[4, N // 2] = 5.0               # single spike: long flat cdf stretches (denom<1e-5 path)
        w[5] = 0.0; w[5, 0] = 1.0; w[5, N - 1] = 1.0  # only the dropped end weights are non-zero
        w[6] = 1.0                                   # exactly uniform
        w[7] = 0.0; w[7, 10] = 0.5; w[7, 11] = 0.5
        pts, tv = ref_utils.sampling_pts_fine_torch(torch.from_numpy(ro), torch.from_numpy(rd),
                                                    torch.from_numpy(ts), torch.from_numpy(w), NF)
        mid = 0.5 * (torch.from_numpy(ts)[..., 1:] + torch.from_numpy(ts)[..., :-1])
        smp = ref_utils.sample_pdf(mid, torch.from_numpy(w)[..., 1:-1], NF, det=True)
        tag = "_%d" % N
        out.update({"rays_o" + tag: ro, "rays_d" + tag: rd, "ts" + tag: ts, "w" + tag: w,
                    "pts" + tag: pts, "tvals" + tag: tv, "samples" + tag: smp})
    save("g6_fine", **out)


# ----------------------------------------------------------------------------- G7 latents + style MLPs
def g7_style():
    out = {}
    lat = ref_models.StyleLatents_variational(style_num=2, frame_num=20, latent_dim=32)
    lat.load_state_dict(t(synth.latents_state(4, style_num=2, frame_num=20)))
    sid = torch.tensor([0, 0, 1, 1, 0, 1, 0, 1], dtype=torch.long)
    fid = torch.tensor([0, 19, 3, 19, 25, 40, 119, 119], dtype=torch.long)   # ids >= frame_num wrap via repeat(7)
    out["style_ids"], out["frame_ids"] = sid, fid
    for sc in (0.0, 1.0, 0.35):
        lat.sigma_scale = sc
        with torch.no_grad():
            out["latents_s%g" % sc] = lat(style_ids=sid, frame_ids=fid, type="llff")

    rng = np.random.default_rng(7)
    M = 24
    x = rng.uniform(-1, 1, (M, 63)).astype(np.float32)
    z = rng.standard_normal((M, 32)).astype(np.float32)
    conc = np.maximum(rng.standard_normal((M, 512)), 0).astype(np.float32)
    cm = ref_models.StyleMLP_before_concat(Args).eval()
    cm.load_state_dict(t(synth.concat_state(2)))
    sm = ref_models.StyleMLP_Wild_multilayers(Args).eval()
    sm.load_state_dict(t(synth.style_state(3)))
    with torch.no_grad():
        cf = cm(x=torch.from_numpy(x), latent=torch.from_numpy(z))["concat_features"]
        rgb = sm(x=torch.from_numpy(x), concated=torch.from_numpy(conc), latent=torch.from_numpy(z))["rgb"]
    out.update(x=x, z=z, conc=conc, concat_features=cf, style_rgb=rgb)
    save("g7_style", **out)


# ----------------------------------------------------------------------------- G8 end-to-end
class _FakeDataset:
    """Just the attributes rendering.cal_geometry reads (rendering.py:9-14)."""

    def __init__(self, h, w, near, far, cps):
        self.mode = "train"
        self.cps = cps
        self.cps_valid = cps
        self.hwf = [h, w, synth.fern_intrinsics(h, w)]
        self.near, self.far = near, far
        self.frame_num, self.h, self.w = cps.shape[0], h, w


class _FakeLoader:
    def __init__(self, dataset, batches):
        self.dataset = dataset
        self._b = batches

    def __iter__(self):
        return iter(self._b)

    def __len__(self):
        return len(self._b)


def g8_end_to_end():
    out = {}
    for nc, nf in ((128, 64), (64, 64)):
        tag = "_%dc%df" % (nc, nf)
        a = type("A", (Args,), {"N_samples": nc, "N_samples_fine": nf})
        h = w = 8
        ro, rd = test_rays(h * w, 800 + nc)
        coarse, fine = make_nerf(0, "coarse", a), make_nerf(1, "fine", a)
        fwd_c = ref_utils.batchify(lambda **kw: coarse(**kw), 32)
        fwd_f = ref_utils.batchify(lambda **kw: fine(**kw), 32)
        ds = _FakeDataset(h, w, 0., 1., np.eye(4, dtype=np.float32)[None])
        batches = [{"rays_o": torch.from_numpy(ro[i:i + 32]), "rays_d": torch.from_numpy(rd[i:i + 32])}
                   for i in range(0, h * w, 32)]
        with tempfile.TemporaryDirectory() as tmp, torch.no_grad():
            rgb_map, t_map = ref_rendering.cal_geometry(
                model_forward=fwd_c, samp_func=ref_utils.sampling_pts_uniform, dataloader=_FakeLoader(ds, batches),
                args=a, device="cpu", sv_path=tmp, model_forward_fine=fwd_f,
                samp_func_fine=ref_utils.sampling_pts_fine_torch)
        out.update({"rays_o" + tag: ro, "rays_d" + tag: rd,
                    "plain_rgb" + tag: rgb_map.reshape(-1, 3), "plain_t" + tag: t_map.reshape(-1)})

        # stylised chain: the reference callables in the order rendering.render_style applies them
        # (rendering.py:118-178); render_style itself only hands back the unconsumed tail, so the
        # per-batch results are collected here.
        coarse.set_enable_style(True)
        fine.set_enable_style(True)
        cm = ref_models.StyleMLP_before_concat(a).eval()
        cm.load_state_dict(t(synth.concat_state(2)))
        sm = ref_models.StyleMLP_Wild_multilayers(a).eval()
        sm.load_state_dict(t(synth.style_state(3)))
        lat = ref_models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
        lat.load_state_dict(t(synth.latents_state(4)))
        lat.sigma_scale = 1.0
        R = h * w
        sid = torch.zeros(R, dtype=torch.long)
        fid = torch.full((R,), 33, dtype=torch.long)
        ro_t, rd_t = torch.from_numpy(ro), torch.from_numpy(rd)
        for jit_tag, seed in (("", None), ("_jit", 4321 + nc)):
            with torch.no_grad():
                if seed is None:
                    pts, ts = ref_utils.sampling_pts_uniform(rays_o=ro_t, rays_d=rd_t, N_samples=nc, near=0., far=1., perturb=False)
                else:
                    torch.manual_seed(seed)
                    pts, ts = ref_utils.sampling_pts_uniform(rays_o=ro_t, rays_d=rd_t, N_samples=nc, near=0., far=1., perturb=True)
                    torch.manual_seed(seed)
                    jit = torch.zeros([R, nc])
                    torch.nn.init.uniform_(jit, 0, 1)
                    out["jit" + tag] = jit
                z = lat(style_ids=sid, frame_ids=fid, type="llff")
                zbar = torch.mean(z, dim=1, keepdims=True)

                def one_pass(model, pts, n):
                    ret = model(pts=pts, dirs=rd_t.unsqueeze(1).expand([R, n, 3]))
                    z1 = z.unsqueeze(1).expand([R, n, 32])
                    cf = cm(x=ret["pts"], latent=z1)["concat_features"]
                    both = torch.concat((ret["base_remap"], cf), dim=-1)
                    zb = torch.unsqueeze(zbar, dim=2).expand([R, n, 32])
                    return sm(x=ret["pts"], concated=both, latent=zb)["rgb"], ret["sigma"]

                rgb_s, sig = one_pass(coarse, pts, nc)
                rc, tc, wc = ref_utils.alpha_composition(rgb_s, sig, ts, 0)
                pts_f, ts_f = ref_utils.sampling_pts_fine_torch(ro_t, rd_t, ts, wc, nf)
                rgb_s, sig = one_pass(fine, pts_f, nc + nf)
                rf, tf, _ = ref_utils.alpha_composition(rgb_s, sig, ts_f, 0)
            out.update({"styled_rgb_coarse" + jit_tag + tag: rc, "styled_rgb" + jit_tag + tag: rf,
                        "styled_t" + jit_tag + tag: tf, "styled_ts_fine" + jit_tag + tag: ts_f})
    save("g8_end_to_end", **out)


# ----------------------------------------------------------------------------- G9 2-D style pass
def g9_style2d():
    out = {}
    rng = np.random.default_rng(9)
    tsd = t(synth.transformer_state(5))
    tr = ref_transformer.Transformer().eval()
    tr.load_state_dict(tsd)

    # nn.MultiheadAttention semantics (the module the reference layers wrap, transformer.py:158)
    q = torch.from_numpy(rng.standard_normal((48, 1, 512)).astype(np.float32))
    k = torch.from_numpy(rng.standard_normal((30, 1, 512)).astype(np.float32))
    v = torch.from_numpy(rng.standard_normal((30, 1, 512)).astype(np.float32))
    with torch.no_grad():
        o = tr.decoder.layers[0].multihead_attn(q, k, v)[0]
    out.update(mha_q=q[:, 0], mha_k=k[:, 0], mha_v=v[:, 0], mha_out=o[:, 0])

    src = torch.from_numpy(rng.standard_normal((48, 1, 512)).astype(np.float32))
    mem = torch.from_numpy(rng.standard_normal((30, 1, 512)).astype(np.float32))
    with torch.no_grad():
        es = tr.encoder_s.layers[1](src, pos=None)
        ec = tr.encoder_c.layers[2](src, pos=src)
        dl = tr.decoder.layers[1](src, mem, pos=None, query_pos=src * 0.5)
    out.update(layer_src=src[:, 0], layer_mem=mem[:, 0], enc_s_out=es[:, 0], enc_c_out=ec[:, 0], declayer_out=dl[:, 0])

    # full Transformer.forward on 6x8 content tokens / 5x7 style tokens (shape bug guard: the output
    # is reshaped with the STYLE map size, so content and style grids must have equal token counts)
    style_map = torch.from_numpy(rng.standard_normal((1, 512, 6, 8)).astype(np.float32))
    content_map = torch.from_numpy(rng.standard_normal((1, 512, 6, 8)).astype(np.float32))
    with torch.no_grad():
        hs = tr(style_map, None, content_map, content_map, None)
    out.update(tr_style=style_map, tr_content=content_map, tr_hs=hs)

    # patch embed, CNN decoder, VGG
    pe = ref_tctrans.PatchEmbed().eval()
    pe.load_state_dict(t(synth.embed_state(6)))
    dec = ref_tctrans.decoder
    dec.load_state_dict(t(synth.decoder_state(7)))
    dec.eval()
    vgg = torch.nn.Sequential(*list(ref_tctrans.vgg.children())[:31])
    vgg.load_state_dict(t(synth.vgg_state(8)))
    vgg.eval()
    img = torch.from_numpy(rng.uniform(0, 1, (1, 3, 21, 29)).astype(np.float32))   # odd -> floor in embed, ceil in pools
    with torch.no_grad():
        emb = pe(img)
        d_in = torch.from_numpy(rng.standard_normal((1, 512, 3, 4)).astype(np.float32))
        d_out = dec(d_in)
    out.update(img=img, embed_out=emb, cnn_in=d_in, cnn_out=d_out)

    net = ref_tctrans.StyTrans(vgg, dec, pe, tr).eval()
    with torch.no_grad():
        feats = net.encode_with_intermediate(img)
    assert torch.equal(feats[3], feats[4])     # enc_5 is empty for vgg[:31] -> identity (tctrans.py:146)
    for i, f in enumerate(feats[:4]):
        out["vgg_%d" % (i + 1)] = f
    m, s = ref_function.calc_mean_std(feats[2])
    m2, s2 = ref_style_function.calc_mean_std(feats[2])
    assert torch.equal(m, m2) and torch.equal(s, s2)
    out.update(ms_mean=m, ms_std=s)
    out["adain"] = ref_style_function.adaptive_instance_normalization(feats[1], feats[1].flip(-1) * 0.5 + 0.1)

    # StyTrans test branch (H != W selects it, tctrans.py:187,233-245) + trans_test post-processing
    content = torch.from_numpy(rng.uniform(0, 1, (1, 3, 40, 56)).astype(np.float32))
    style = torch.from_numpy(synth.style_image(11, 40, 56))
    with torch.no_grad():
        ics, hs2 = net(content, style)
        up = torch.nn.Upsample(size=(40, 56), mode="bilinear", align_corners=True)(ics)   # trans_test.py:172-173
        rows = hs2.reshape(-1, 512)                                                        # trans_test.py:176
        feat = torch.cat([rows.mean(dim=0), rows.var(dim=0)])[None]
    out.update(st_content=content, st_hs=hs2, st_ics=ics, st_image=up, st_feature=feat)
    save("g9_style2d", **out)


# ----------------------------------------------------------------------------- G10 image epilogue
def g10_image():
    """What rendering.cal_geometry hands to imageio.imwrite for each finished frame (rendering.py:66-73): the inert
    imageio placeholder is given a recording imwrite for the duration of the call."""
    import imageio  # the placeholder registered by ref_shim.install()
    a = type("A", (Args,), {"N_samples": 64, "N_samples_fine": 64})
    h, w, frames = 6, 8, 3
    ro, rd = test_rays(frames * h * w, 1010)
    coarse, fine = make_nerf(0, "coarse", a), make_nerf(1, "fine", a)
    fwd_c = ref_utils.batchify(lambda **kw: coarse(**kw), 32)
    fwd_f = ref_utils.batchify(lambda **kw: fine(**kw), 32)
    ds = _FakeDataset(h, w, 0., 1., np.tile(np.eye(4, dtype=np.float32)[None], (frames, 1, 1)))
    batches = [{"rays_o": torch.from_numpy(ro[i:i + 40]), "rays_d": torch.from_numpy(rd[i:i + 40])}
               for i in range(0, frames * h * w, 40)]   # 40 does not divide a frame: frames complete mid-batch
    written = []
    imageio.imwrite = lambda path, arr: written.append((os.path.basename(path), np.array(arr)))
    try:
        with tempfile.TemporaryDirectory() as tmp, torch.no_grad():
            rgb_map, t_map = ref_rendering.cal_geometry(
                model_forward=fwd_c, samp_func=ref_utils.sampling_pts_uniform, dataloader=_FakeLoader(ds, batches),
                args=a, device="cpu", sv_path=tmp, model_forward_fine=fwd_f,
                samp_func_fine=ref_utils.sampling_pts_fine_torch)
    finally:
        del imageio.imwrite
    names = [n for n, _ in written]
    assert names == [x for i in range(frames) for x in ("rgb_%05d.png" % i, "depth_%05d.png" % i)], names
    save("g10_image", rgb=rgb_map.reshape(-1, 3), t=t_map.reshape(-1), frames=np.int64(frames), h=np.int64(h), w=np.int64(w),
         rgb8=np.stack([arr for n, arr in written if n.startswith("rgb")]),
         depth8=np.stack([arr for n, arr in written if n.startswith("depth")]))


# ----------------------------------------------------------------------------- G11 LLFF poses
def g11_llff_poses():
    """load_llff.load_llff_data (load_llff.py:233-305) run for real on a synthetic scene directory: a seeded
    poses_bounds.npy plus empty image files, with the inert imageio placeholder returning blank frames of the right
    size.  Records the raw array and everything the loader derives from it (recentred poses, bounds, the 120-view
    spiral, the hold-out index) and dataset.py:101-102's cps_valid."""
    import imageio
    import load_llff as ref_llff
    rng = np.random.default_rng(11)
    n, H0, W0, focal0, factor = 20, 96, 128, 110.0, 8
    arr = np.zeros((n, 17))
    for i in range(n):
        # LLFF convention (before the axis fix-up of load_llff.py:239): columns [down, right, backwards], position, hwf
        ang = rng.normal(0, 0.08, 3)
        cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
        R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
             @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
        pose = np.concatenate([R, rng.normal(0, 0.6, (3, 1)), np.array([[H0], [W0], [focal0]])], 1)
        arr[i, :15] = pose.reshape(-1)
        arr[i, 15:] = [rng.uniform(1.6, 2.4), rng.uniform(14, 22)]
    out = {"poses_arr": arr, "factor": np.int64(factor), "image_hw": np.array([H0 // factor, W0 // factor])}
    blank = np.zeros((H0 // factor, W0 // factor, 3), np.uint8)
    imageio.imread = lambda f: np.zeros((H0, W0, 3), np.uint8) if os.sep + "images" + os.sep in f else blank
    try:
        with tempfile.TemporaryDirectory() as tmp:
            np.save(os.path.join(tmp, "poses_bounds.npy"), arr)
            for d in ("images", "images_%d" % factor):
                os.makedirs(os.path.join(tmp, d))
                for i in range(n):
                    open(os.path.join(tmp, d, "%03d.png" % i), "wb").close()
            for flat in (False,):   # path_zflat=True divides N_views into a float and np.linspace refuses it (load_llff.py:283)
                images, poses, bds, render_poses, i_test = ref_llff.load_llff_data(tmp, factor, recenter=True, bd_factor=.75,
                                                                                   spherify=False, path_zflat=flat)
                tag = ""
                cps_valid = np.concatenate([render_poses[:, :3, :4], np.zeros_like(render_poses[:, :1, :4])], axis=1)  # dataset.py:101
                cps_valid[:, 3, 3] = 1.
                out.update({"poses" + tag: poses, "bds" + tag: bds, "render_poses" + tag: render_poses,
                            "i_test" + tag: np.int64(i_test), "cps_valid" + tag: cps_valid})
            assert images.shape == (n, H0 // factor, W0 // factor, 3)
    finally:
        del imageio.imread
    save("g11_llff_poses", **out)


def g12_vae():
    """VAE.encode on three style-feature rows (train_tgtcs.py:148-155: the latent table's mu / logvar when no latent
    checkpoint exists), and the shapes set_latents produces."""
    vae = ref_models.VAE(data_dim=1024, latent_dim=32, W=512, D=4, kl_lambda=0.1).eval()
    vae.load_state_dict(t(synth.vae_state(9)))
    rng = np.random.default_rng(12)
    feats = np.concatenate([rng.standard_normal((3, 512)) * 0.5, np.abs(rng.standard_normal((3, 512))) * 0.3], 1).astype(np.float32)
    with torch.no_grad():
        _, mu, logvar = vae.encode(torch.from_numpy(feats), various=False)
    lat = ref_models.StyleLatents_variational(style_num=3, frame_num=5, latent_dim=32)
    lat.style_latents_mu = torch.nn.Parameter(mu.detach())
    lat.style_latents_logvar = torch.nn.Parameter(logvar.detach())
    lat.set_latents()
    save("g12_vae", style_features=feats, mu=mu, logvar=logvar, latents_shape=np.array(lat.latents.shape),
         state_keys=np.array(sorted(vae.state_dict().keys())))


# ----------------------------------------------------------------------------- G13 files on disk
def g13_files():
    """Files as the REFERENCE writes them, kept as fixtures under tests/golden/files/ (a few KB each):

    * geometry_00001.npz and geometry.npz -- written by the reference's own np.savez calls (rendering.py:73 and :81) during
      a cal_geometry run on 3 frames of 6x8 pixels (the run of g10, inputs regenerable from seeds);
    * 000500.tar, style_120500.tar, latent_120500.tar -- the reference's torch.save sites sit inside the closures of
      train() (train_tgtcs.py:284-300, :503-517) and cannot be reached without a training run, so these hold what those
      sites hold: the state dicts of the reference's OWN modules (width 8, so the files stay small) and of the optimisers
      built the way train_tgtcs.py:37,54 builds them (after one step, so that they carry state), under the sites' keys;
    * stylized_data.npz -- trans_test.py:179's keyword layout (a dict, a str, two float32 arrays) with the shapes of
      trans_test.py:145,176-178; the driver around it needs torchvision and checkpoint files that do not exist offline.
    """
    import imageio
    out_dir = os.path.join(HERE, "files")
    os.makedirs(out_dir, exist_ok=True)
    a = type("A", (Args,), {"N_samples": 64, "N_samples_fine": 64})
    h, w, frames = 6, 8, 3
    ro, rd = test_rays(frames * h * w, 1010)
    coarse, fine = make_nerf(0, "coarse", a), make_nerf(1, "fine", a)
    cps = np.tile(np.eye(4, dtype=np.float32)[None], (frames, 1, 1))
    cps[:, 0, 3] = np.arange(frames)
    ds = _FakeDataset(h, w, 0., 1., cps)
    batches = [{"rays_o": torch.from_numpy(ro[i:i + 40]), "rays_d": torch.from_numpy(rd[i:i + 40])} for i in range(0, frames * h * w, 40)]
    imageio.imwrite = lambda path, arr: None
    try:
        with tempfile.TemporaryDirectory() as tmp, torch.no_grad():
            ref_rendering.cal_geometry(
                model_forward=ref_utils.batchify(lambda **kw: coarse(**kw), 32), samp_func=ref_utils.sampling_pts_uniform,
                dataloader=_FakeLoader(ds, batches), args=a, device="cpu", sv_path=tmp,
                model_forward_fine=ref_utils.batchify(lambda **kw: fine(**kw), 32), samp_func_fine=ref_utils.sampling_pts_fine_torch)
            for name in ("geometry_00001.npz", "geometry.npz"):
                with open(os.path.join(tmp, name), "rb") as f, open(os.path.join(out_dir, name), "wb") as g:
                    g.write(f.read())
    finally:
        del imageio.imwrite

    small = type("S", (Args,), {"netwidth": 8, "netwidth_fine": 8, "lrate": 5e-4})
    torch.manual_seed(13)
    model, model_fine = ref_models.StyleNerf(args=small, mode='coarse'), ref_models.StyleNerf(args=small, mode='fine')
    concat_model, style_model = ref_models.StyleMLP_before_concat(small), ref_models.StyleMLP_Wild_multilayers(small)
    latents = ref_models.StyleLatents_variational(style_num=2, frame_num=3, latent_dim=32)
    optimizer = torch.optim.Adam(params=list(model.parameters()) + list(model_fine.parameters()), lr=small.lrate, betas=(0.9, 0.999))
    style_optimizer = torch.optim.Adam(params=list(style_model.parameters()) + list(concat_model.parameters()), lr=small.lrate, betas=(0.9, 0.999))
    for opt, mods in ((optimizer, (model, model_fine)), (style_optimizer, (style_model, concat_model))):
        sum((p ** 2).sum() for m in mods for p in m.parameters()).backward()
        opt.step()
    torch.save({'global_step': 500, 'model': model.state_dict(), 'model_fine': model_fine.state_dict(),
                'optimizer': optimizer.state_dict(), 'style_optimizer': style_optimizer.state_dict()}, os.path.join(out_dir, "000500.tar"))
    torch.save({'global_step': 120500, 'model': style_model.state_dict(), 'concat_model': concat_model.state_dict(),
                'optimizer': style_optimizer.state_dict()}, os.path.join(out_dir, "style_120500.tar"))
    torch.save({'global_step': 120500, 'train_set_1': latents.state_dict()}, os.path.join(out_dir, "latent_120500.tar"))

    rng = np.random.default_rng(13)
    style_feature = np.zeros([1, 1024], dtype=np.float32)
    for _ in range(2):      # trans_test.py:176-178: one row appended per frame, then the mean over the appended rows
        style_feature = np.append(style_feature, [rng.standard_normal(1024).astype(np.float32)], axis=0)
    style_feature = np.sum(style_feature, axis=0, keepdims=True) / (style_feature.shape[0] - 1)
    np.savez(os.path.join(out_dir, 'stylized_data'), style_names={"starry_night": 0}, style_paths="./style/starry_night.jpg",
             style_images=rng.random((1, 16, 16, 3)).astype(np.float32), style_features=style_feature)
    for name in sorted(os.listdir(out_dir)):
        print("files/%-22s %8.1f KB" % (name, os.path.getsize(os.path.join(out_dir, name)) / 1024))


# ----------------------------------------------------------------------------- G14 training step
def g14_train():
    """One iteration of the reference's `Origin_train` body (train_tgtcs.py:226-254) and the coherence term of `Style_train`
    (train_tgtcs.py:394-403, :451-458 with VGGNet.py:204-210 and utils.py L2_norm), run through the REFERENCE's own functions
    with autograd: losses, composited colours and the gradients of all 2 x 24 parameters.

    The body draws three random tensors from torch's global generator -- the stratified jitter inside sampling_pts_uniform
    (utils.py:519-520) and the density noise inside each alpha_composition (utils.py:371-374).  They are reproduced here by
    drawing the same three tensors, in the same order, from the same seed, and stored as inputs."""
    import VGGNet as ref_vggnet
    R, N, NF, std = 96, 32, 32, 1.0
    ro, rd = test_rays(R, 1414)
    rng = np.random.default_rng(1415)
    gt = rng.uniform(0, 1, (R, 3)).astype(np.float32)

    class A(Args):
        N_samples, N_samples_fine, sigma_noise_std = N, NF, std
    model, model_fine = make_nerf(0, "coarse", A).train(), make_nerf(1, "fine", A).train()
    seed = 141414
    torch.manual_seed(seed)
    jit = torch.zeros(R, N)
    torch.nn.init.uniform_(jit, 0, 1)
    noise_c = torch.randn(R, N) * std
    noise_f = torch.randn(R, N + NF) * std
    torch.manual_seed(seed)                                   # the body itself, as train() runs it
    rays_o, rays_d, rgb_gt = torch.from_numpy(ro), torch.from_numpy(rd), torch.from_numpy(gt)
    pts, ts = ref_utils.sampling_pts_uniform(rays_o=rays_o, rays_d=rays_d, N_samples=N, near=0., far=1., perturb=True)
    ret = model(pts=pts, dirs=rays_d.unsqueeze(1).expand([R, N, 3]))
    rgb_exp, t_exp, weights = ref_utils.alpha_composition(ret['rgb'], ret['sigma'], ts, std)
    loss_rgb = ref_utils.img2mse(rgb_gt, rgb_exp)
    pts_fine, ts_fine = ref_utils.sampling_pts_fine_torch(rays_o, rays_d, ts, weights, NF)
    ret_f = model_fine(pts=pts_fine, dirs=rays_d.unsqueeze(1).expand([R, N + NF, 3]))
    rgb_exp_fine, _, _ = ref_utils.alpha_composition(ret_f['rgb'], ret_f['sigma'], ts_fine, std)
    loss_rgb_fine = ref_utils.img2mse(rgb_gt, rgb_exp_fine)
    loss = loss_rgb + loss_rgb_fine
    loss.backward()
    out = dict(rays_o=ro, rays_d=rd, rgb_gt=gt, jitter=jit, noise_coarse=noise_c, noise_fine=noise_f, n_coarse=N, n_fine=NF,
               sigma_noise_std=std, seeds=np.array([0, 1]), ts=ts, ts_fine=ts_fine, rgb_exp=rgb_exp, rgb_exp_fine=rgb_exp_fine,
               loss_rgb=loss_rgb, loss_rgb_fine=loss_rgb_fine, loss=loss)
    for tag, m in (("coarse", model), ("fine", model_fine)):
        for k, p in m.state_dict(keep_vars=True).items():
            out["grad_%s.%s" % (tag, k)] = p.grad
    # the coherence term: two consecutive frame-ordered batches of R2 rays (cnt = 1: neither the first batch nor a restart)
    R2 = 64
    x_prev = torch.from_numpy(rng.uniform(0, 1, (R2, 3)).astype(np.float32))            # rgb_exp_style2 of the previous batch
    y_prev = torch.from_numpy(rng.uniform(0, 1, (R2, 3)).astype(np.float32))            # rgb_exp_style_fine2 of the previous batch
    xo_prev = torch.from_numpy(rng.uniform(0, 1, (R2, 3)).astype(np.float32))           # rgb_origin2 of the previous batch
    rgb2 = torch.from_numpy(rng.uniform(0, 1, (R2, 3)).astype(np.float32)).requires_grad_()
    rgb_fine2 = torch.from_numpy(rng.uniform(0, 1, (R2, 3)).astype(np.float32)).requires_grad_()
    rgb_origin2 = torch.from_numpy(rng.uniform(0, 1, (R2, 3)).astype(np.float32))
    loss_coh = ref_utils.L2_norm(ref_vggnet.cosine_similarity(rgb2, x_prev) - ref_vggnet.cosine_similarity(rgb_origin2, xo_prev))
    x_origin = rgb_origin2                                                              # :402-403 run before the fine term
    loss_coh = loss_coh + ref_utils.L2_norm(ref_vggnet.cosine_similarity(rgb_fine2, y_prev)
                                            - ref_vggnet.cosine_similarity(rgb_origin2, x_origin))
    loss_coh.backward()
    out.update(coh_x_prev=x_prev, coh_y_prev=y_prev, coh_x_origin_prev=xo_prev, coh_rgb2=rgb2, coh_rgb_fine2=rgb_fine2,
               coh_rgb_origin2=rgb_origin2, coh_loss=loss_coh, coh_grad_rgb2=rgb2.grad, coh_grad_rgb_fine2=rgb_fine2.grad)
    save("g14_train", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1_rays, g2_coarse, g3_embed, g4_nerf, g5_composite, g6_fine, g7_style, g8_end_to_end, g9_style2d, g10_image,
               g11_llff_poses, g12_vae, g13_files, g14_train):
        if not only or fn.__name__.split("_")[0] in only:
            fn()
