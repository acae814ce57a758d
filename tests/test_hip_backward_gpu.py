"""GPU: training-side gradients.  The differentiable layer-by-layer NeRF network on the HIP GEMM
kernel (autograd_ops.py: forward tgtc_s2d_linear, backward tgtc_s2d_linear_backward / tgtc_s2d_activation) through the HIP
compositing backward, against float64 autograd on the CPU oracle; and one `Origin_train`-shaped optimisation step
(train_tgtcs.py:218-262)."""
import numpy as np
import pytest
import torch

from oracle import fields, raymarch
from tgtc_style_amd import synth

pytestmark = pytest.mark.gpu


def T(sd, dtype=None):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) if dtype else torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


class Args:
    use_viewdir, act_type, embed_freq_coor, embed_freq_dir = True, "relu", 10, 4
    netdepth = netdepth_fine = 8
    netwidth = netwidth_fine = 256
    style_D, vae_latent, precision = 8, 32, "fp16x3"


@pytest.mark.parametrize("M,K,N,relu", [(1, 5, 3, False), (37, 63, 256, True), (1000, 319, 256, True), (4097, 256, 1, False), (20000, 256, 128, True)])
def test_linear_backward_matches_float64(M, K, N, relu):
    from tgtc_style_amd import autograd_ops as ao
    rng = np.random.default_rng(M + K)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    lin = torch.nn.Linear(K, N)
    g = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    xc = x.cuda().requires_grad_()
    lc = torch.nn.Linear(K, N).cuda()
    lc.load_state_dict(lin.state_dict())
    y = ao.linear(xc, lc, relu)
    (y * g.cuda()).sum().backward()
    # float64 reference; the ReLU gate is taken from the HIP forward (a pre-activation within 1e-7 of zero may fall on
    # either side, and among millions of outputs a few do)
    xd, wd, bd = x.double().requires_grad_(), lin.weight.detach().double().requires_grad_(), lin.bias.detach().double().requires_grad_()
    yd = torch.nn.functional.linear(xd, wd, bd)
    yd = yd * (y.detach().cpu() > 0).double() if relu else yd
    (yd * g.double()).sum().backward()
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(y.detach(), yd.detach()) <= 2e-6
    assert rel(xc.grad, xd.grad) <= 5e-6 and rel(lc.weight.grad, wd.grad) <= 5e-6 and rel(lc.bias.grad, bd.grad) <= 5e-6


@pytest.mark.parametrize("fused", [True, False])
def test_nerf_gradients_match_the_oracle(fused):
    """d loss / d every weight of the coarse NeRF through sampling (no gradient), the network, compositing with the density
    regulariser, and an MSE loss -- HIP kernels vs float64 autograd on the oracle's formulas.  fused: the fused training path
    (csrc/mlp_train.hip: forward with stash, input-gradient chain, weight-gradient kernel); not fused: one GEMM per product."""
    from tgtc_style_amd import models, utils
    rng = np.random.default_rng(0)
    R, N = 96, 64
    ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3)))
    rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1.0])
    jit = torch.from_numpy(rng.uniform(0, 1, (R, N)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((R, N)).astype(np.float32))
    gt = torch.from_numpy(rng.uniform(0, 1, (R, 3)).astype(np.float32))
    sd = synth.nerf_state(0)
    # the oracle's formulas with autograd, in float64 and -- as the yardstick of what float32 arithmetic does to these
    # gradients (ReLU gates of pre-activations within rounding of zero fall on either side) -- in float32
    def oracle_grads(dtype):
        w = {k: v.clone().to(dtype).requires_grad_() for k, v in T(sd).items()}
        pts_o, ts_o = raymarch.sample_coarse(ro, rd, N, 0., 1., jitter=jit)
        pe = fields.posenc(pts_o, 10).to(dtype).reshape(R * N, -1)
        de = fields.posenc(rd[:, None, :].expand(R, N, 3), 4).to(dtype).reshape(R * N, -1)
        ret = fields.nerf_mlp(w, pe, de)
        rgb_o, _, _ = raymarch.composite(ret["rgb"].reshape(R, N, 3), ret["sigma"].reshape(R, N) + noise.to(dtype), ts_o.to(dtype))
        loss = ((rgb_o - gt.to(dtype)) ** 2).mean()
        loss.backward()
        return float(loss.detach()), {k: v.grad.double() for k, v in w.items()}
    loss_o, g64 = oracle_grads(torch.float64)
    _, g32 = oracle_grads(torch.float32)
    # HIP
    m = models.StyleNerf(Args, mode="coarse")
    m.load_state_dict(T(sd))
    m = m.cuda().trainable(fused=fused)
    pts, ts = utils.sampling_pts_uniform(ro.cuda(), rd.cuda(), N_samples=N, near=0., far=1., jitter=jit.cuda())
    out = m(pts=pts, dirs=rd.cuda()[:, None, :].expand(R, N, 3))
    rgb, _, _ = utils.alpha_composition(out["rgb"], out["sigma"], ts, noise=noise.cuda())
    loss = ((rgb - gt.cuda()) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - loss_o) <= 1e-5 * abs(loss_o)
    worst, bad = 0.0, []
    for k, p in m.state_dict(keep_vars=True).items():
        g_ref = g64[k]
        assert p.grad is not None, k
        scale = float(g_ref.abs().max()) + 1e-30
        err = float((p.grad.double().cpu() - g_ref).abs().max()) / scale
        yard = float((g32[k] - g_ref).abs().max()) / scale
        worst = max(worst, err)
        print("%-32s HIP vs f64 %.2e   torch-f32 vs f64 %.2e" % (k, err, yard))
        bad = bad + [(k, err, yard)] if err > max(2e-5, 3 * yard) else bad
    print("worst relative gradient error", worst)
    assert not bad, bad
    # the fused forward-only path is what a network that is not marked trainable keeps using, grad mode or not
    m.trainable(False)
    out2 = m(pts=pts, dirs=rd.cuda()[:, None, :].expand(R, N, 3))
    assert not out2["rgb"].requires_grad
    assert float((out2["rgb"] - out["rgb"].detach()).abs().max()) <= 2e-5


@pytest.mark.parametrize("fused", [True, False])
def test_origin_train_step_reduces_the_loss(fused):
    """A few iterations of the reference's Origin_train body (coarse + fine losses, sigma noise, Adam) on a fixed batch."""
    from tgtc_style_amd import models, utils
    rng = np.random.default_rng(1)
    R = 256
    ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3))).cuda()
    rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1.0]).cuda()
    gt = torch.from_numpy(rng.uniform(0.2, 0.8, (R, 3)).astype(np.float32)).cuda()
    model, model_fine = models.StyleNerf(Args, mode="coarse"), models.StyleNerf(Args, mode="fine")
    model.load_state_dict(T(synth.nerf_state(0))), model_fine.load_state_dict(T(synth.nerf_state(1)))
    model, model_fine = model.cuda().trainable(fused=fused), model_fine.cuda().trainable(fused=fused)
    opt = torch.optim.Adam(list(model.parameters()) + list(model_fine.parameters()), lr=5e-4)
    from tgtc_style_amd import training
    gen = torch.Generator(device="cuda").manual_seed(0)
    losses = []
    for it in range(6):
        r = training.origin_train_step(model, model_fine, opt, ro, rd, gt, 64, 64, 0., 1., sigma_noise_std=0.1,
                                       jitter=torch.rand(R, 64, device="cuda", generator=gen))
        assert set(r) == {"loss", "loss_rgb", "loss_rgb_fine"}
        losses.append(r["loss"])
    print(losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    if fused:
        for net in (model, model_fine):
            net._trainer.status()          # no scaled gradient left the fp16 operand range


def test_fused_training_path_matches_the_unfused_one():
    """The fused training path against the layer-by-layer one on the SAME inputs, ragged sample counts (M not a multiple of the
    128-sample workgroup tile, nor of the 32-sample weight-gradient step): outputs and every one of the 24 gradients."""
    from tgtc_style_amd import fused_train, models
    for M in (1, 100, 1000, 4097):
        rng = np.random.default_rng(M)
        pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (M, 3))).cuda()
        dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3))).cuda()
        g_rgb = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32) * 1e-3).cuda()
        g_sig = torch.from_numpy(rng.standard_normal(M).astype(np.float32) * 1e-5).cuda()
        grads = {}
        for fused in (True, False):
            m = models.StyleNerf(Args, mode="fine")
            m.load_state_dict(T(synth.nerf_state(1)))
            m = m.cuda().trainable(fused=fused)
            out = m(pts=pts, dirs=dirs)
            ((out["rgb"] * g_rgb).sum() + (out["sigma"] * g_sig).sum()).backward()
            grads[fused] = (out["rgb"].detach(), out["sigma"].detach(), [p.grad.clone() for p in fused_train.mlp_parameters(m.net)])
            if fused:
                m._trainer.status()
        a, b = grads[True], grads[False]
        assert float((a[0] - b[0]).abs().max()) <= 2e-5 and float((a[1] - b[1]).abs().max()) <= 2e-5 * float(b[1].abs().max())
        # The two forwards round differently at the 1e-7 level: among 4 097 x 2 432 ReLU units a few pre-activations within
        # rounding of zero gate differently.  Such a sample s changes its own pre-activation gradients in every layer below the
        # flipped unit, i.e. it adds a RANK-ONE term d_dz[s] (x) h[s] to each weight gradient (dense in the early layers) and
        # d_dz[s] to the bias gradient.  So per layer: dW_fused - dW_unfused is at most eight rank-one terms (eight flipped
        # samples; four seen at M = 4 097: singular values 3e-4 .. 9e-5, the fifth 1e-7) on top of 3e-5 of rounding, the bias
        # difference lies in the span of the same left singular vectors, and both are small (a sample is 1/sqrt(M) of a sum).
        for l in range(12):
            dW, db = (a[2][2 * l] - b[2][2 * l]).double(), (a[2][2 * l + 1] - b[2][2 * l + 1]).double()
            sW, sb = float(b[2][2 * l].abs().max()) + 1e-30, float(b[2][2 * l + 1].abs().max()) + 1e-30
            assert float(dW.abs().max()) <= 5e-2 * sW and float(db.abs().max()) <= 5e-2 * sb, (M, l)
            U, S, Vh = torch.linalg.svd(dW, full_matrices=False)
            r = min(8, S.numel())
            dW = dW - (U[:, :r] * S[:r]) @ Vh[:r]
            db = db - U[:, :r] @ (U[:, :r].T @ db)
            assert float(dW.abs().max()) <= 3e-5 * sW, (M, l, "weight", float(dW.abs().max()) / sW, S[:10].tolist())
            assert float(db.abs().max()) <= 3e-5 * sb, (M, l, "bias", float(db.abs().max()) / sb, S[:10].tolist())


def _float64_preactivations(sd, pts, dirs):
    """every ReLU layer's pre-activations of the oracle's network in float64: [M, units] (models.py:95-111)"""
    w = {k: v.double() for k, v in T(sd).items()}
    pe, de = fields.posenc(pts, 10).double(), fields.posenc(dirs, 4).double()
    lin = lambda name, x: x @ w["net.%s.weight" % name].T + w["net.%s.bias" % name]
    zs, h = [], None
    for i in range(8):
        x = pe if i == 0 else (torch.cat([pe, h], -1) if i == 5 else h)
        z = lin("base_layers.%d" % i, x)
        zs.append(z)
        h = torch.relu(z)
    z = lin("base_remap_layer", h)
    zs.append(z)
    z = lin("rgb_layers.0", torch.cat([torch.relu(z), de], -1))
    zs.append(z)
    return torch.cat(zs, -1)


def test_fused_and_unfused_gradients_agree_element_wise_away_from_relu_kinks():
    """the comparison above projects rank-one terms away, which a wrong-row bug in the weight gradients would survive
    too.  The direct form: drop the samples that have ANY pre-activation within 1e-4 of zero in the float64 oracle (the only
    ones whose ReLU gates two float32 forwards can disagree on), run both paths on the rest, and compare every gradient element
    by element at rounding level -- no projection."""
    from tgtc_style_amd import fused_train, models
    M0 = 4097
    rng = np.random.default_rng(M0)
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (M0, 3)))
    dirs = torch.from_numpy(rng.uniform(-1, 1, (M0, 3)))
    sd = synth.nerf_state(1)
    z = _float64_preactivations(sd, pts, dirs)
    keep = (z.abs() >= 1e-4).all(-1)
    if int(keep.sum()) % 32 == 0:                 # stay ragged against the 32-sample weight-gradient step
        keep[int(torch.nonzero(keep)[-1])] = False
    M = int(keep.sum())
    print("%d of %d samples have every pre-activation at least 1e-4 from zero" % (M, M0))
    assert M >= 0.6 * M0
    pts, dirs = pts[keep].cuda(), dirs[keep].cuda()
    g_rgb = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32) * 1e-3).cuda()
    g_sig = torch.from_numpy(rng.standard_normal(M).astype(np.float32) * 1e-5).cuda()
    grads = {}
    for fused in (True, False):
        m = models.StyleNerf(Args, mode="fine")
        m.load_state_dict(T(sd))
        m = m.cuda().trainable(fused=fused)
        out = m(pts=pts, dirs=dirs)
        ((out["rgb"] * g_rgb).sum() + (out["sigma"] * g_sig).sum()).backward()
        grads[fused] = [p.grad.clone() for p in fused_train.mlp_parameters(m.net)]
    worst = 0.0
    for i, (a, b) in enumerate(zip(grads[True], grads[False])):
        e = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        worst = max(worst, e)
        assert e <= 1e-5, (i, e)          # measured 1.4e-6 on 2 992 of 4 097 samples
    print("worst element-wise difference of the 24 gradients, relative to each tensor's maximum: %.2e" % worst)


@pytest.mark.parametrize("fused", [True, False])
def test_origin_train_step_matches_the_reference_golden(golden, fused):
    """g14 (tests/golden/gen_golden.py g14_train) = one Origin_train iteration run by the REFERENCE's own code and autograd
    (train_tgtcs.py:226-254) with its three random draws recorded.  The HIP training path on the same inputs: the two
    losses, the composited colours and all 2 x 24 parameter gradients against the reference's, under the rule of the oracle
    test above: max(2e-5, 3 x what float32 autograd itself is away from float64) relative to the tensor maximum."""
    import test_oracle_golden as og
    from tgtc_style_amd import models, utils
    g = golden("g14_train")
    R, N, NF = g["rays_o"].shape[0], int(g["n_coarse"]), int(g["n_fine"])
    ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
    gt = torch.from_numpy(g["rgb_gt"]).cuda()
    nets = []
    for seed, mode in zip(g["seeds"], ("coarse", "fine")):
        m = models.StyleNerf(Args, mode=mode)
        m.load_state_dict(T(synth.nerf_state(int(seed))))
        nets.append(m.cuda().trainable(fused=fused))
    pts, ts = utils.sampling_pts_uniform(ro, rd, N_samples=N, near=0., far=1., jitter=torch.from_numpy(g["jitter"]).cuda())
    ret = nets[0](pts=pts, dirs=rd[:, None, :].expand(R, N, 3))
    rgb_c, _, w_c = utils.alpha_composition(ret["rgb"], ret["sigma"], ts, noise=torch.from_numpy(g["noise_coarse"]).cuda())
    # The fine depths come out of the (non-differentiated, utils.py:562-579) inverse-CDF sampler, whose ill-conditioned bins
    # turn the 2e-6 by which two correct coarse passes differ into a different depth (tests/conditioning.py): the HIP sampler
    # must agree with the reference's on all but a handful of samples, and the fine half of the step -- where the gradients
    # are -- is then compared on the REFERENCE's depths, so that both backward passes differentiate the same function.
    _, ts_hip = utils.sampling_pts_fine_torch(ro, rd, ts, w_c.detach(), NF)
    dts = (ts_hip.cpu() - torch.from_numpy(g["ts_fine"])).abs()
    print("fine depths: %d of %d samples differ from the reference's by more than 2e-5 (max %.2e)" % (int((dts > 2e-5).sum()), dts.numel(), float(dts.max())))
    assert float((dts > 2e-5).float().mean()) <= 1e-2      # (a moved sample shifts its neighbours' places in the sorted merge)
    ts_f = torch.from_numpy(g["ts_fine"]).cuda()
    pts_f = ro[:, None, :] + rd[:, None, :] * ts_f[..., None].double()          # utils.py:578
    ret = nets[1](pts=pts_f, dirs=rd[:, None, :].expand(R, N + NF, 3))
    rgb_f, _, _ = utils.alpha_composition(ret["rgb"], ret["sigma"], ts_f, noise=torch.from_numpy(g["noise_fine"]).cuda())
    l_c, l_f = ((rgb_c - gt) ** 2).mean(), ((rgb_f - gt) ** 2).mean()
    (l_c + l_f).backward()
    assert float((ts.cpu() - torch.from_numpy(g["ts"])).abs().max()) <= 1e-7
    assert float((rgb_c.detach().cpu() - torch.from_numpy(g["rgb_exp"])).abs().max()) <= 2e-5
    assert float((rgb_f.detach().cpu() - torch.from_numpy(g["rgb_exp_fine"])).abs().max()) <= 2e-5
    assert abs(float(l_c.detach()) - float(g["loss_rgb"])) <= 1e-5 * float(g["loss_rgb"])
    assert abs(float(l_f.detach()) - float(g["loss_rgb_fine"])) <= 1e-5 * float(g["loss_rgb_fine"])
    o64 = og._oracle_train_step(g, torch.float64, golden_depths=True)          # the yardstick: how far the reference's float32 autograd is from float64
    worst, bad = 0.0, []
    for tag, m, g64 in zip(("coarse", "fine"), nets, o64["grads"]):
        for k, p in m.state_dict(keep_vars=True).items():
            ref = torch.from_numpy(g["grad_%s.%s" % (tag, k)]).double()
            scale = float(ref.abs().max()) + 1e-30
            err = float((p.grad.double().cpu() - ref).abs().max()) / scale
            yard = float((g64[k].double() - ref).abs().max()) / scale
            worst = max(worst, err)
            if err > max(2e-5, 3 * yard):
                bad.append((tag, k, err, yard))
    print("HIP (%s) vs the reference's own autograd: worst relative gradient difference %.2e" % ("fused" if fused else "per layer", worst))
    assert not bad, bad
    if fused:
        assert nets[0].training_overflows() == 0 and nets[1].training_overflows() == 0


def test_two_forwards_before_one_backward_keep_their_own_stash():
    """the reference's Origin_train calls the net through utils.batchify (utils.py:435-456), i.e. TWO forwards of
    32 768 samples before ONE backward; the fused path's activation stash must belong to the call, not to the trainer.  The
    chunked forward + one backward equals the one-call forward + backward, and equals the per-layer path's chunked run."""
    from tgtc_style_amd import fused_train, models, utils
    M, chunk = 3000, 1024                     # three chunks, the last one ragged
    rng = np.random.default_rng(77)
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (M, 3))).cuda()
    dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3))).cuda()
    g_rgb = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32) * 1e-3).cuda()
    g_sig = torch.from_numpy(rng.standard_normal(M).astype(np.float32) * 1e-5).cuda()
    res = {}
    for tag, fused, chunked in (("fused-chunked", True, True), ("fused-whole", True, False), ("layers-chunked", False, True)):
        m = models.StyleNerf(Args, mode="fine")
        m.load_state_dict(T(synth.nerf_state(1)))
        m = m.cuda().trainable(fused=fused)
        fwd = utils.batchify(lambda **k: m(**k), chunk) if chunked else (lambda **k: m(**k))
        out = fwd(pts=pts, dirs=dirs)
        ((out["rgb"] * g_rgb).sum() + (out["sigma"] * g_sig).sum()).backward()
        res[tag] = (out["rgb"].detach(), out["sigma"].detach(), [p.grad.clone() for p in fused_train.mlp_parameters(m.net)])
        if fused:
            m._trainer.status()
            assert m.training_overflows() == 0
    a, b, c = res["fused-chunked"], res["fused-whole"], res["layers-chunked"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                    # the forward does not depend on the chunking
    for l in range(24):
        scale = float(b[2][l].abs().max()) + 1e-30
        # same samples, same gates (bit-identical forwards): only the order of the fp32 atomics in the weight gradients differs
        assert float((a[2][l] - b[2][l]).abs().max()) <= 2e-5 * scale, ("chunked vs whole", l, float((a[2][l] - b[2][l]).abs().max()) / scale)
        # and the per-layer path on the same chunks agrees up to the few rank-one gate-flip terms of the test above: 5 % raw
        assert float((a[2][l] - c[2][l]).abs().max()) <= 5e-2 * (float(c[2][l].abs().max()) + 1e-30), ("fused vs layers", l)
    # with a stash shared between calls the first chunk's gradients would have been computed from the LAST chunk's activations:
    # show that this test can see it -- the gradient of the first chunk alone is far from that of the last chunk alone
    m = models.StyleNerf(Args, mode="fine")
    m.load_state_dict(T(synth.nerf_state(1)))
    m = m.cuda().trainable()
    parts = []
    for lo in (0, 2 * chunk):
        m.zero_grad()
        out = m(pts=pts[lo:lo + chunk], dirs=dirs[lo:lo + chunk])
        ((out["rgb"] * g_rgb[lo:lo + chunk]).sum() + (out["sigma"] * g_sig[lo:lo + chunk]).sum()).backward()
        parts.append(fused_train.mlp_parameters(m.net)[2].grad.clone())
    assert float((parts[0] - parts[1]).abs().max()) > 0.2 * float(parts[0].abs().max())


@pytest.mark.parametrize("s_rgb,s_sigma", [(1.0, 1.0), (1e-3, 1e3), (1e3, 1e-3), (1e-6, 1.0), (1.0, 0.0)])
def test_fused_backward_takes_head_gradients_of_any_ratio(s_rgb, s_sigma):
    """The sigma row joins the input-gradient chain two layers after the colour head (mlp_train.hip D2): whatever the ratio
    of the two upstream gradients, the chain's scales must hold both.  (Round 4 found NaN weight gradients for
    |d sigma| ~ |d rgb| ~ 1: the colour branch had shrunk, the lagged scale had grown, and d sigma left the fp16 range.)
    Upstream gradients given directly: loss = sum(w_rgb * rgb) + sum(w_sigma * sigma), fused kernels vs float64 autograd."""
    from tgtc_style_amd import fused_train, models
    M = 512
    rng = np.random.default_rng(11)
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (M, 3)))
    dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3)))
    w_rgb = torch.from_numpy(rng.standard_normal((M, 3)) * s_rgb)
    w_sig = torch.from_numpy(rng.standard_normal((M,)) * s_sigma)
    sd = synth.nerf_state(1)

    def oracle_grads(dtype):
        w = {k: v.clone().to(dtype).requires_grad_() for k, v in T(sd).items()}
        ret = fields.nerf_mlp(w, fields.posenc(pts, 10).to(dtype), fields.posenc(dirs, 4).to(dtype))
        ((ret["rgb"] * w_rgb.to(dtype)).sum() + (ret["sigma"].reshape(M) * w_sig.to(dtype)).sum()).backward()
        return {k: v.grad.double() for k, v in w.items()}
    g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    m = models.StyleNerf(Args, mode="fine")
    m.load_state_dict(T(sd))
    m = m.cuda().trainable()
    before = m.training_overflows()
    out = m(pts=pts.cuda(), dirs=dirs.cuda())
    ((out["rgb"] * w_rgb.cuda().float()).sum() + (out["sigma"].reshape(M) * w_sig.cuda().float()).sum()).backward()
    m._trainer.status()
    assert m.training_overflows() == before
    bad = []
    for k, p in m.state_dict(keep_vars=True).items():
        scale = float(g64[k].abs().max()) + 1e-30
        err = float((p.grad.double().cpu() - g64[k]).abs().max()) / scale
        yard = float((g32[k] - g64[k]).abs().max()) / scale
        print("%-32s HIP vs f64 %.2e   torch-f32 vs f64 %.2e" % (k, err, yard))
        bad = bad + [(k, err, yard)] if not err <= max(2e-5, 3 * yard) else bad
    assert not bad, bad


def test_overflow_guard_zero_fills_the_gradients():
    """growth above ~2^7 in one transposed layer overflows the fp16 operands of the input-gradient chain (and a
    forward that overflows gives NaN gradients); an Adam step on them would destroy the weights.  The library's guard
    zero-fills non-finite or overflowed gradients on the device and counts the event; a normal backward afterwards works and
    leaves the count alone."""
    from tgtc_style_amd import fused_train, models
    M = 512
    rng = np.random.default_rng(5)
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (M, 3))).cuda()
    dirs = torch.from_numpy(rng.uniform(-1, 1, (M, 3))).cuda()
    m = models.StyleNerf(Args, mode="fine")
    m = m.cuda().trainable()
    fired = []
    # (a) the backward's own range: layer 7 shrinks the gradient by 2^-k, layer 6 grows it back by 2^k -- the forward stays in
    # range, the chain's scale (one layer behind) does not; (b) a forward that leaves the fp16 range outright
    cases = [("chain 2^%d" % k, {"net.base_layers.6.weight": 2.0 ** k, "net.base_layers.7.weight": 2.0 ** -k}) for k in (10, 14, 18)]
    cases.append(("forward", {"net.base_layers.3.weight": 1e6}))
    for name, scale in cases:
        sd = {k: v.copy() * np.float32(scale.get(k, 1.0)) for k, v in synth.nerf_state(1).items()}
        m.load_state_dict({k: v.cuda() for k, v in T(sd).items()})
        m.zero_grad()
        before = m.training_overflows()
        out = m(pts=pts, dirs=dirs)
        (out["rgb"].sum() + out["sigma"].sum()).backward()
        hit = m.training_overflows() - before
        grads = [p.grad for p in fused_train.mlp_parameters(m.net)]
        assert all(bool(torch.isfinite(g).all()) for g in grads), name          # never a non-finite gradient, fired or not
        if hit:
            fired.append(name)
            assert hit == 1 and all(float(g.abs().max()) == 0.0 for g in grads), name
            with pytest.raises(RuntimeError, match="fp16 range"):
                m._trainer.status()
        print(name, "-> guard fired" if hit else "-> in range")
    assert "forward" in fired and any(n.startswith("chain") for n in fired), fired
    # a well-scaled network on the same trainer: finite, non-zero gradients, counter unchanged
    before = m.training_overflows()
    m.load_state_dict({k: v.cuda() for k, v in T(synth.nerf_state(1)).items()})
    m.zero_grad()
    out = m(pts=pts, dirs=dirs)
    (out["rgb"].sum() + out["sigma"].sum()).backward()
    m._trainer.status()
    assert m.training_overflows() == before
    assert float(fused_train.mlp_parameters(m.net)[0].grad.abs().max()) > 0


def test_style_mlp_gradients_match_the_oracle():
    """The two style MLPs marked trainable: gradients w.r.t. every weight, the latent and the concat features against
    float64 autograd on the oracle (Style_train's differentiable pieces, train_tgtcs.py:312-571)."""
    from tgtc_style_amd import models
    rng = np.random.default_rng(3)
    M = 700
    x = torch.from_numpy(rng.uniform(-1, 1, (M, 63)).astype(np.float32))
    z = torch.from_numpy(rng.standard_normal((M, 32)).astype(np.float32))
    remap = torch.from_numpy(np.maximum(rng.standard_normal((M, 256)), 0).astype(np.float32))
    gt = torch.from_numpy(rng.uniform(0, 1, (M, 3)).astype(np.float32))
    csd, ssd = synth.concat_state(2), synth.style_state(3)

    cm, sm = models.StyleMLP_before_concat(Args), models.StyleMLP_Wild_multilayers(Args)
    cm.load_state_dict(T(csd)), sm.load_state_dict(T(ssd))
    cm, sm = cm.cuda().trainable(), sm.cuda().trainable()
    cm.act_trace, sm.act_trace = [], []
    zc = z.cuda().requires_grad_()
    cf = cm(x=x.cuda(), latent=zc)["concat_features"]
    rgb = sm(x=x.cuda(), concated=torch.cat([remap.cuda(), cf], -1), latent=zc)["rgb"]
    ((rgb - gt.cuda()) ** 2).mean().backward()
    c_gates = [(h > 0).cpu() for h in cm.act_trace]
    s_gates = [(h > 0).cpu() for h in sm.act_trace]
    assert len(c_gates) == 5 and len(s_gates) == 7

    def oracle(dtype):
        """The oracle's formulas (fields.concat_mlp / style_mlp) with each ReLU gate taken from the HIP forward: with 700
        samples one pre-activation within rounding of zero falling on the other side would move a gradient by 1e-3."""
        cw = {k: v.clone().to(dtype).requires_grad_() for k, v in T(csd).items()}
        sw = {k: v.clone().to(dtype).requires_grad_() for k, v in T(ssd).items()}
        zz = z.clone().to(dtype).requires_grad_()
        xx = x.to(dtype)
        lin = lambda w, i, h: torch.nn.functional.linear(h, w["layers.%d.weight" % i], w["layers.%d.bias" % i])
        h = xx
        for i in range(5):
            h = torch.cat([h, zz], -1)
            h = torch.cat([h, xx], -1) if i == 4 else h
            h = lin(cw, i, h) * c_gates[i].to(dtype)
        h = torch.cat([remap.to(dtype), h, xx], -1)
        for i in range(7):
            h = torch.cat([h, zz], -1)
            h = torch.cat([h, xx], -1) if i == 4 else h
            h = lin(sw, i, h) * s_gates[i].to(dtype)
        rgb = torch.sigmoid(lin(sw, 7, torch.cat([h, zz], -1)))
        ((rgb - gt.to(dtype)) ** 2).mean().backward()
        g = {"c." + k: v.grad.double() for k, v in cw.items()}
        g.update({"s." + k: v.grad.double() for k, v in sw.items()})
        g["z"] = zz.grad.double()
        return g, rgb.detach()
    (g64, rgb64), (g32, _) = oracle(torch.float64), oracle(torch.float32)
    assert float((rgb.detach().double().cpu() - rgb64).abs().max()) <= 2e-5
    # the gated formulas are the oracle's: same outputs as fields.concat_mlp / fields.style_mlp
    ref_cf = fields.concat_mlp(T(csd), x, z)["concat_features"]
    ref_rgb = fields.style_mlp(T(ssd), x, torch.cat([remap, ref_cf], -1), z)["rgb"]
    assert float((ref_rgb.double() - rgb64).abs().max()) <= 2e-5
    got = {"c." + k: p.grad for k, p in cm.state_dict(keep_vars=True).items()}
    got.update({"s." + k: p.grad for k, p in sm.state_dict(keep_vars=True).items()})
    got["z"] = zc.grad
    bad = []
    for k, ref in g64.items():
        scale = float(ref.abs().max()) + 1e-30
        err = float((got[k].double().cpu() - ref).abs().max()) / scale
        yard = float((g32[k] - ref).abs().max()) / scale
        if err > max(2e-5, 3 * yard):
            bad.append((k, err, yard))
    assert not bad, bad
    # not trainable: the packed kernels, no graph
    out = sm.trainable(False)(x=x.cuda(), concated=torch.cat([remap.cuda(), cf.detach()], -1), latent=z.cuda())["rgb"]
    assert not out.requires_grad and float((out - rgb.detach()).abs().max()) <= 2e-5


def test_latent_table_gradients():
    """StyleLatents_variational marked trainable: d/d latents and d/d mu of the gathered rows (models.py:490-506) against
    autograd on the oracle's formula, with repeated rows and the x7 wrap."""
    from tgtc_style_amd import models
    sd = synth.latents_state(4, style_num=2, frame_num=20)
    sid = torch.tensor([0, 0, 1, 1, 0, 1, 0, 1, 0, 0], dtype=torch.long)
    fid = torch.tensor([0, 19, 3, 19, 25, 40, 119, 119, 0, 19], dtype=torch.long)
    g = torch.from_numpy(np.random.default_rng(5).standard_normal((10, 32)).astype(np.float32))
    for scale in (0.0, 0.35, 1.0):
        w = {k: v.clone().double().requires_grad_() for k, v in T(sd).items()}
        ref = fields.latents_forward(w, sid, fid, sigma_scale=scale, llff=True)
        (ref * g.double()).sum().backward()
        lat = models.StyleLatents_variational(style_num=2, frame_num=20, latent_dim=32)
        lat.load_state_dict(T(sd))
        lat = lat.cuda().trainable()
        lat.sigma_scale = scale
        out = lat(style_ids=sid.cuda(), frame_ids=fid.cuda(), type="llff")
        (out * g.cuda()).sum().backward()
        assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-6
        assert float((lat.latents.grad.cpu().double() - w["latents"].grad).abs().max()) <= 1e-6
        assert float((lat.style_latents_mu.grad.cpu().double() - w["style_latents_mu"].grad).abs().max()) <= 1e-6


def test_style_train_step_reduces_the_loss():
    """The rendering / pixel / -log p part of a Style_train iteration: the style MLPs and the latent table learn (their
    parameters move, the loss falls) while the NeRF networks stay frozen on their fused kernels."""
    from tgtc_style_amd import models, training
    rng = np.random.default_rng(2)
    R = 200
    ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (R, 3))).cuda()
    rd = torch.from_numpy(rng.uniform(-1, 1, (R, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1.0]).cuda()
    gt = torch.from_numpy(rng.uniform(0.2, 0.8, (R, 3)).astype(np.float32)).cuda()
    sid = torch.zeros(R, dtype=torch.long).cuda()
    fid = torch.from_numpy(rng.integers(0, 20, R)).cuda()
    model, model_fine = models.StyleNerf(Args, mode="coarse"), models.StyleNerf(Args, mode="fine")
    model.load_state_dict(T(synth.nerf_state(0))), model_fine.load_state_dict(T(synth.nerf_state(1)))
    model, model_fine = model.cuda(), model_fine.cuda()
    model.set_enable_style(True), model_fine.set_enable_style(True)
    cm, sm = models.StyleMLP_before_concat(Args), models.StyleMLP_Wild_multilayers(Args)
    cm.load_state_dict(T(synth.concat_state(2))), sm.load_state_dict(T(synth.style_state(3)))
    lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    lat.load_state_dict(T(synth.latents_state(4, style_num=1, frame_num=20)))
    cm, sm, lat = cm.cuda().trainable(), sm.cuda().trainable(), lat.cuda().trainable()
    lat.sigma_scale = 1.0
    before = [p.detach().clone() for p in (cm.layers[0].weight, sm.layers[7].weight, lat.latents)]
    nerf_before = model.net.base_layers[0].weight.detach().clone()
    opt = torch.optim.Adam(list(cm.parameters()) + list(sm.parameters()) + [lat.latents], lr=1e-3)
    gen = torch.Generator(device="cuda").manual_seed(0)
    losses = [training.style_train_step(model, model_fine, cm, sm, lat, opt, ro, rd, gt, sid, fid, 64, 64, 0., 1., sigma_noise_std=0.1,
                                        logp_loss_lambda=1e-3, jitter=torch.rand(R, 64, device="cuda", generator=gen))["loss"]
              for _ in range(6)]
    print(losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    after = (cm.layers[0].weight, sm.layers[7].weight, lat.latents)
    assert all(float((a.detach() - b).abs().max()) > 0 for a, b in zip(after, before))
    assert torch.equal(model.net.base_layers[0].weight.detach(), nerf_before) and model.net.base_layers[0].weight.grad is None


def test_style_train_coherence_term_matches_the_oracle():
    """The second, frame-ordered batch of a Style_train iteration and its cosine-coherence term (train_tgtcs.py:366-403,
    :436-458; VGGNet.py:204-210; utils.py:459): two consecutive iterations through training.style_train_step.  The
    coherence loss of the second one and its gradient w.r.t. the latent table -- which reaches the loss through the latent
    gather, both style MLPs and compositing of the coarse AND the fine pass -- against float64 autograd on the oracle's
    formulas (carried colours detached, as CoherenceState documents)."""
    from tgtc_style_amd import models, training
    rng = np.random.default_rng(7)
    R, NC, NF = 48, 64, 64
    sds = [synth.nerf_state(0), synth.nerf_state(1), synth.concat_state(2), synth.style_state(3)]
    lsd = synth.latents_state(4, style_num=1, frame_num=20)

    def batch(seed):
        g = np.random.default_rng(seed)
        return {"rays_o": torch.from_numpy(g.uniform(-0.3, 0.3, (R, 3))), "rays_d": torch.from_numpy(g.uniform(-1, 1, (R, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1.0]),
                "rgb_origin": torch.from_numpy(g.uniform(0.1, 0.9, (R, 3)).astype(np.float32)), "style_id": torch.zeros(R, dtype=torch.long),
                "frame_id": torch.full((R,), seed % 20, dtype=torch.long), "jitter": torch.from_numpy(g.uniform(0, 1, (R, NC)).astype(np.float32))}
    b1, b2, main = batch(11), batch(12), batch(13)
    gt = torch.from_numpy(rng.uniform(0.2, 0.8, (R, 3)).astype(np.float32))

    model, model_fine = models.StyleNerf(Args, mode="coarse"), models.StyleNerf(Args, mode="fine")
    model.load_state_dict(T(sds[0])), model_fine.load_state_dict(T(sds[1]))
    model, model_fine = model.cuda(), model_fine.cuda()
    model.set_enable_style(True), model_fine.set_enable_style(True)
    cm, sm = models.StyleMLP_before_concat(Args), models.StyleMLP_Wild_multilayers(Args)
    cm.load_state_dict(T(sds[2])), sm.load_state_dict(T(sds[3]))
    lat = models.StyleLatents_variational(style_num=1, frame_num=20, latent_dim=32)
    lat.load_state_dict(T(lsd))
    cm, sm, lat = cm.cuda().trainable(), sm.cuda().trainable(), lat.cuda().trainable()
    lat.sigma_scale = 1.0
    opt = torch.optim.SGD(list(cm.parameters()) + list(sm.parameters()) + [lat.latents], lr=0.0)    # lr 0: weights stay the oracle's
    cuda = lambda b: {k: v.cuda() for k, v in b.items()}
    state = training.CoherenceState(frame_num=20)
    kw = dict(sigma_noise_std=0.0, rgb_loss_lambda=0.0, logp_loss_lambda=0.0, coherence=state, loss_coh_lambda=1.0)
    args = (model, model_fine, cm, sm, lat, opt, main["rays_o"].cuda(), main["rays_d"].cuda(), gt.cuda(), main["style_id"].cuda(),
            main["frame_id"].cuda(), NC, NF, 0., 1.)
    r1 = training.style_train_step(*args, jitter=main["jitter"].cuda(), coh_batch=cuda(b1), **kw)
    assert r1["loss_coh"] == 0.0 and state.cnt == 1                      # first batch: nothing to compare with (:400)
    r2 = training.style_train_step(*args, jitter=main["jitter"].cuda(), coh_batch=cuda(b2), **kw)
    got_grad = lat.latents.grad.detach().cpu().double()                  # rgb / logp weights are 0: this is d loss_coh / d latents
    assert state.cnt == 2 and r2["loss_coh"] > 0

    def oracle(dtype):
        w = [T(sd, dtype) for sd in sds]
        lw = {k: v.clone().to(dtype).requires_grad_() for k, v in T(lsd).items()}

        def styled(b):
            ro, rd = b["rays_o"], b["rays_d"]
            z = fields.latents_forward(lw, b["style_id"], b["frame_id"], sigma_scale=1.0, llff=True)
            pts, ts = raymarch.sample_coarse(ro, rd, NC, 0., 1., b["jitter"].to(dtype), dtype=dtype)
            rgb, sig = fields._styled_pass(w[0], w[2], w[3], pts, rd[:, None, :].expand(-1, NC, -1), z, dtype)
            rgb_c, _, wc = raymarch.composite(rgb, sig, ts)
            pts_f, ts_f = raymarch.sample_fine(ro, rd, ts, wc.detach(), NF)
            rgb, sig = fields._styled_pass(w[1], w[2], w[3], pts_f, rd[:, None, :].expand(-1, NC + NF, -1), z, dtype)
            rgb_f, _, _ = raymarch.composite(rgb, sig, ts_f)
            return rgb_c, rgb_f
        cos = lambda a, b: ((a / (a.norm(dim=1, keepdim=True) + 1e-8)) * (b / (b.norm(dim=1, keepdim=True) + 1e-8))).sum(1)
        l2 = lambda x: torch.sqrt((x ** 2).sum() + 1e-8)
        with torch.no_grad():
            x, y = styled(b1)
        c, f = styled(b2)
        o1, o2 = b1["rgb_origin"].to(dtype), b2["rgb_origin"].to(dtype)
        loss = l2(cos(c, x) - cos(o2, o1)) + l2(cos(f, y) - cos(o2, o2))      # :401, :456 (x_origin is already o2 there)
        loss.backward()
        return float(loss), lw["latents"].grad.double()
    loss64, grad64 = oracle(torch.float64)
    loss32, grad32 = oracle(torch.float32)
    print("loss_coh %.6f (oracle f64 %.6f, f32 %.6f)" % (r2["loss_coh"], loss64, loss32))
    assert abs(r2["loss_coh"] - loss64) <= 1e-3 * max(1.0, abs(loss64))
    scale = float(grad64.abs().max())
    err, yard = float((got_grad - grad64).abs().max()) / scale, float((grad32 - grad64).abs().max()) / scale
    print("d loss_coh / d latents: rel err %.2e (float32 oracle: %.2e)" % (err, yard))
    assert scale > 0 and err <= max(2e-3, 5 * yard)
