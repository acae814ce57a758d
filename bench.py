#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the plain NeRF render (128 coarse + 64 fine samples per ray) on synthetic
fern-shaped 400x400 frames (BASELINE.json config 2), one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--sharding frames|rays]

A "step" of the headline = every rank renders ONE whole 400x400 frame (160 000 rays) of its own camera pose:
on-device ray generation (+NDC warp), then the library's render of those rays (tgtc_render_rays_plain: coarse depths ->
PE + coarse NeRF MLP -> weights -> inverse-CDF fine sampling -> PE + fine NeRF MLP -> alpha compositing) -- for the default
precision pair (coarse fp16x3 + fine fp16mx) the split path: seven launches, of which the fine pass (mlp_nerf_mx2.hip, the
two-tile persistent kernel) and the coarse pass are 99.9 % of the time; for the other pairs ONE launch of the fused ray kernel
(render_fused.hip: a wavefront owns a ray, per-sample tensors never exist) -- then (N > 1) one RCCL all-gather of the
[160000, 4] RGB+depth images.
Per-GPU work is fixed as N grows (weak scaling); `value` = N * 160000 * steps / max-over-ranks wall time.
`--sharding rays` (BASELINE config 4, strong scaling): every step renders ONE frame, each rank a contiguous 1/N of
its rays (parallel.shard_range), reassembled by the same all-gather; `value` = 160000 * steps / time.
Inputs resident in HBM when the timed region starts: packed weights, camera poses.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel, timed live with HIP events on the launch stream: split path = the fine pass's kernel on the
                frame's ray count (launched again after the timed region); fused path = the ray kernel around every timed step.
                `whole_render` prices all launches of a frame together, as rounds 1-3 priced the fused kernel
  cpu_baseline  the CPU oracle on the host cores (N = 1 only; bounded sample)
  configs       N = 1 only: BASELINE configs 3 and 4 on this GPU -- the stylised ray path, the 2-D style pass,
                and a whole 504x378 trex frame -- each with its own live-timed value
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 400
N_COARSE, N_FINE = 128, 64
MAC_FULL = 593408          # MACs per sample, full NeRF
MAC_SIGMA = 491264         # MACs per sample, trunk + sigma head
MAC_STYLED = 1506912       # stylised fine sample: trunk + sigma + remap + concat MLP + style MLP
FLOP_PER_RAY = 2 * (N_COARSE * MAC_SIGMA + (N_COARSE + N_FINE) * MAC_FULL)        # 353.6 MFLOP
FLOP_PER_RAY_STYLED = 2 * (N_COARSE * MAC_SIGMA + (N_COARSE + N_FINE) * MAC_STYLED)   # 704.4 MFLOP
PEAK_FP16_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
BYTES_PER_RAY = 2 * 3 * 8 + 16   # float64 origin + direction in, RGB + depth out (SURVEY 8d counts float32 rays: 24 + 16)
BYTES_PER_RAY_STYLED = BYTES_PER_RAY + 32 * 4   # + the ray's 32-float style latent
# MFMA issue slots per algorithmic product: fp16x3 = hi*hi + lo*hi + hi*lo; fp16mx = 4 f16 + 2 fp6 16x16x128 per 128-deep block
MFMA_PER_PRODUCT = {"fp16x3": 3.0, "fp16": 1.0, "fp16mx": 1.5}
DTYPE = {"fp16x3": "f16 MFMA operands split hi+lo (3 products), f32 accumulate",
         "fp16": "f16 MFMA operands, f32 accumulate",
         "fp16mx": "f16 MFMA product + two block-scaled fp6 (e2m3) correction products, f32 accumulate"}
NOTES = {"fp16": "single fp16 MFMA product: ~1e-3 per-network error, outside the 1e-3 north-star tolerance end to end",
         "fp16x3": "fp16 hi+lo split, three MFMA products in both passes: fp32-equivalent (rgb 2e-5 against the reference's renders)",
         "fp16x3+fp16mx": "coarse pass fp16x3, fine pass fp16 product + two block-scaled fp6 correction products: rgb 2.5e-4, "
                          "depth 2.1e-4 against the reference's own renders, every 1e-3 parity test green "
                          "(the inverse-CDF step amplifies coarse-pass errors only, so the coarse pass stays fp16x3)"}


class NetArgs:
    use_viewdir, act_type = True, "relu"
    embed_freq_coor, embed_freq_dir = 10, 4
    netdepth = netdepth_fine = 8
    netwidth = netwidth_fine = 256
    style_D, vae_latent = 8, 32
    precision = "fp16x3"


def t_state(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def build_nets(precision):
    """`precision` = one mode for both networks, or "coarse+fine" (precision belongs to each network handle)."""
    from tgtc_style_amd import models, synth
    nets = []
    per_net = precision.split("+") * 2
    for (seed, mode), prec in zip(((0, "coarse"), (1, "fine")), per_net):
        a = type("A", (NetArgs,), {"precision": prec})
        m = models.StyleNerf(a, mode=mode)
        m.load_state_dict(t_state(synth.nerf_state(seed)))
        m = m.cuda()
        m.packed()
        nets.append(m)
    return nets


def mfma_per_product(precision):
    """Weighted over the coarse (sigma-only) and fine (full) passes of one ray."""
    pc, pf = (precision.split("+") * 2)[:2]
    fc, ff = N_COARSE * MAC_SIGMA, (N_COARSE + N_FINE) * MAC_FULL
    return (MFMA_PER_PRODUCT[pc] * fc + MFMA_PER_PRODUCT[pf] * ff) / (fc + ff)


def cpu_budget():
    """CPUs this job may actually use: its affinity mask, capped by the cgroup's CPU quota where one is set -- a GPU box hands a
    one-GPU job 16 CPUs' worth of a 256-CPU host through the quota, not through the mask (256 threads on that share ran the
    baseline 15x slower than 16).  Returns (threads, how they were chosen)."""
    n = len(os.sched_getaffinity(0))
    how = "the job's affinity mask (%d CPUs)" % n
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                q = max(1, int(round(int(quota) / int(period))))
                if q < n:
                    n, how = q, "the cgroup CPU quota (%s / %s) inside an affinity mask of %d" % (quota, period, len(os.sched_getaffinity(0)))
            break
        except (OSError, ValueError, IndexError):
            continue
    if n > 32:      # no quota visible on a many-core host: the pool's documented share for a one-GPU job
        n, how = 16, "16: the GPU pool's CPU share of a one-GPU job (no cgroup quota visible; affinity mask %d CPUs)" % len(os.sched_getaffinity(0))
    return n, how


def cpu_baseline(n_rays, chunk=1024, budget_s=25.0):
    """The CPU oracle (plain PyTorch restatement of the reference, pinned to the reference's goldens) on a
    bounded sample of the same workload; same chunking as the reference CLI (--chunk 1024).  Stops after `budget_s` seconds
    of timed work if the sample has not been rendered by then (the rate is per ray either way)."""
    from oracle import fields, rays
    from tgtc_style_amd import synth
    cores, how = cpu_budget()
    torch.set_num_threads(cores)
    c, f = t_state(synth.nerf_state(0)), t_state(synth.nerf_state(1))
    o, d = rays.frame_rays_ndc(H, W, synth.fern_intrinsics(H, W), synth.spiral_pose(0))
    o, d = torch.from_numpy(o[:n_rays + chunk]), torch.from_numpy(d[:n_rays + chunk])
    done = 0
    with torch.no_grad():
        fields.render_plain(c, f, o[:chunk], d[:chunk], N_COARSE, N_FINE)       # warm-up chunk
        t0 = time.perf_counter()
        for lo in range(chunk, chunk + n_rays, chunk):
            fields.render_plain(c, f, o[lo:lo + chunk], d[lo:lo + chunk], N_COARSE, N_FINE)
            done += min(chunk, chunk + n_rays - lo)
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "rays/s", "cores": cores, "kind": "port", "host_cpus": os.cpu_count(),
            "sample": "%d rays of the same 400x400 frame in chunks of %d after one warm-up chunk, torch %s CPU fp32, "
                      "%d threads = %s (the host has %s CPUs), %.1f s" % (done, chunk, torch.__version__, cores, how, os.cpu_count(), dt)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under profiles/ (bench.py cannot collect
    counters itself): returns (bytes or None, provenance)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        e = table["kernels"][kernel]
        return e["fetch_bytes"] + e["write_bytes"], "profiles/pmc_traffic.json: %s (%s, commit %s)" % (
            e["source"], table.get("collected", "?"), table.get("commit", "?"))
    except (OSError, KeyError, ValueError):
        return None, "no PMC profile committed for this kernel"


class Timed:
    """HIP events on the launch stream around every timed step (the ops of this package launch on torch's current
    stream, so torch.cuda.Event brackets them); the mean is the device time of what a step enqueues."""

    def __init__(self, steps):
        self.ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def run(self, i, fn):
        self.ev[i][0].record()
        out = fn()
        self.ev[i][1].record()
        return out

    def mean_ms(self):
        return float(np.mean([a.elapsed_time(b) for a, b in self.ev]))


def time_loop(fn, steps, warmup):
    """fn(step) -> output; returns (wall seconds of `steps` steps, mean device ms of what fn enqueues, last output)."""
    for i in range(warmup):
        out = fn(i)
    torch.cuda.synchronize()
    timed = Timed(steps)
    t0 = time.perf_counter()
    for i in range(steps):
        out = timed.run(i, lambda: fn(warmup + i))
    torch.cuda.synchronize()
    return time.perf_counter() - t0, timed.mean_ms(), out


def bench_style2d(precision, steps, warmup):
    """Config 3's per-frame 2-D pass at 400x400: patch-embed both images, transformer, CNN decoder, resize, style
    feature, plus one VGG encode (what train-time consumers read).  ms/frame."""
    from tgtc_style_amd import style2d, synth
    mods = {}
    for name, cls, sd in (("tr", style2d.Transformer, synth.transformer_state(5)), ("pe", style2d.PatchEmbed, synth.embed_state(6)),
                          ("dec", style2d.Decoder, synth.decoder_state(7)), ("vgg", style2d.VGG, synth.vgg_state(8))):
        m = cls()
        m.load_state_dict(t_state(sd))
        m.precision = precision
        mods[name] = m.cuda()
    net = style2d.StyTrans(mods["vgg"], mods["dec"], mods["pe"], mods["tr"])
    content = torch.rand(1, 3, H, W, device="cuda")
    style = torch.from_numpy(synth.style_image(11, H, W)).cuda()
    parts = {}
    for label, fn in (("stylize (embed x2 + transformer + decoder + resize + feature)", lambda i: style2d.stylize_frame(net, content, style)),
                      ("vgg encode_with_intermediate", lambda i: net.encode_with_intermediate(content))):
        parts[label] = time_loop(fn, steps, warmup)[1]
    flop = 0.5e12   # ~0.5 TFLOP per 400x400 frame (transformer 165 GMAC, decoder 38.6, VGG 38.6, embeds)
    ms = sum(parts.values())
    return {"metric": "ms/frame, 2-D style pass at 400x400 (2500 tokens): ViT + CNN decoder + VGG", "value": ms, "unit": "ms",
            "higher_is_better": False, "steps": steps, "dtype": precision, "parts_ms": parts,
            "roofline": {"bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop / (ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, "kernel_ms": ms,
                         "note": "all kernels of the pass together (HIP events around the frame)"}}


# one Origin_train iteration: MACs per network sample, forward / weight gradient (the same products) / input gradient (only
# the activation columns of a layer carry a gradient: not the 63 encoding columns of L0 and L5, not the 27 of rgb_layers.0)
MAC_TRAIN_DGRAD = MAC_FULL - 63 * 256 - 63 * 256 - 27 * 128
FLOP_TRAIN_PER_SAMPLE = 2 * (2 * MAC_FULL + MAC_TRAIN_DGRAD)      # 3.489 MFLOP


class _TorchNerf(torch.nn.Module):
    """MLP_style / StyleNerf (reference models.py:63-117, :182-223) as stock torch modules: the comparator of `train_step`."""

    def __init__(self, sd):
        super().__init__()
        lin = lambda name: self._linear(sd["net." + name + ".weight"], sd["net." + name + ".bias"])
        self.base = torch.nn.ModuleList([lin("base_layers.%d" % i) for i in range(8)])
        self.sigma, self.remap = lin("sigma_layer"), lin("base_remap_layer")
        self.rgb = torch.nn.ModuleList([lin("rgb_layers.0"), lin("rgb_layers.1")])

    @staticmethod
    def _linear(w, b):
        m = torch.nn.Linear(w.shape[1], w.shape[0])
        m.weight.data.copy_(torch.from_numpy(w)), m.bias.data.copy_(torch.from_numpy(b))
        return m

    @staticmethod
    def embed(x, L):
        out = [x]
        for k in range(L):
            out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
        return torch.cat(out, -1).to(torch.float32)

    def forward(self, pts, dirs):
        pe, de = self.embed(pts, 10), self.embed(dirs, 4)
        h = torch.relu(self.base[0](pe))
        for i in range(7):
            if i == 4:
                h = torch.cat([pe, h], -1)
            h = torch.relu(self.base[i + 1](h))
        sigma = self.sigma(h).squeeze(-1)
        f = torch.relu(self.rgb[0](torch.cat([torch.relu(self.remap(h)), de], -1)))
        return torch.sigmoid(self.rgb[1](f)), sigma


def _torch_origin_train_step(m, mf, opt, ro, rd, gt, nc, nf, noise_std):
    """train_tgtcs.py:226-254 in stock torch-ROCm eager autograd (fp32 rocBLAS GEMMs): the comparator, not a target."""
    R = ro.shape[0]
    ts = torch.linspace(0., 1., nc, device=ro.device).expand(R, nc)
    mid = .5 * (ts[..., 1:] + ts[..., :-1])
    ts = torch.cat([ts[..., :1], mid], -1) + (torch.cat([mid, ts[..., -1:]], -1) - torch.cat([ts[..., :1], mid], -1)) * torch.rand_like(ts)

    def composite(rgb, sigma, t):
        delta = torch.cat([t[..., 1:] - t[..., :-1], torch.full_like(t[..., :1], 1e10)], -1)
        alpha = 1. - torch.exp(-torch.relu(torch.relu(sigma + torch.randn_like(sigma) * noise_std)) * delta)
        trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
        w = alpha * trans
        return (w[..., None] * rgb).sum(-2), w

    rgb, sigma = m(ro[:, None, :] + ts[..., None].double() * rd[:, None, :], rd[:, None, :].expand(R, nc, 3))
    rgb_c, w = composite(rgb, sigma, ts)
    with torch.no_grad():                                   # sample_pdf on detached weights (utils.py:573-609)
        pdf = w[:, 1:-1] + 1e-5
        pdf = pdf / pdf.sum(-1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
        u = torch.linspace(0., 1., nf, device=ro.device).expand(R, nf).contiguous()
        idx = torch.searchsorted(cdf, u, right=True)
        lo, hi = (idx - 1).clamp(min=0), idx.clamp(max=cdf.shape[-1] - 1)
        c0, c1, b0, b1 = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi), torch.gather(mid, -1, lo), torch.gather(mid, -1, hi)
        den = torch.where(c1 - c0 < 1e-5, torch.ones_like(c0), c1 - c0)
        tf = torch.sort(torch.cat([ts, b0 + (u - c0) / den * (b1 - b0)], -1), -1)[0]
    rgb, sigma = mf(ro[:, None, :] + tf[..., None].double() * rd[:, None, :], rd[:, None, :].expand(R, nc + nf, 3))
    rgb_f, _ = composite(rgb, sigma, tf)
    loss = torch.mean((rgb_c - gt) ** 2) + torch.mean((rgb_f - gt) ** 2)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


def bench_train_step(steps, warmup, rays=1024):
    """one iteration of the reference's Origin_train body (coarse + fine losses, sigma noise, Adam) on
    the FUSED training kernels (fused_train.NerfTrainer: one forward kernel with the activation stash, one input-gradient
    chain, one weight-gradient kernel per network), with the same iteration on the unfused per-layer HIP dense layers and in
    stock torch eager autograd beside it.  ms per iteration."""
    from tgtc_style_amd import models, synth, training
    rng = np.random.default_rng(1)
    ro = torch.from_numpy(rng.uniform(-0.3, 0.3, (rays, 3))).cuda()
    rd = torch.from_numpy(rng.uniform(-1, 1, (rays, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1.0]).cuda()
    gt = torch.from_numpy(rng.uniform(0.2, 0.8, (rays, 3)).astype(np.float32)).cuda()

    def timed(fn):
        for i in range(warmup + steps):
            if i == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, r

    def hip_iteration(fused):
        m, mf = models.StyleNerf(NetArgs, mode="coarse"), models.StyleNerf(NetArgs, mode="fine")
        m.load_state_dict(t_state(synth.nerf_state(0))), mf.load_state_dict(t_state(synth.nerf_state(1)))
        m, mf = m.cuda().trainable(fused=fused), mf.cuda().trainable(fused=fused)
        opt = torch.optim.Adam(list(m.parameters()) + list(mf.parameters()), lr=5e-4)
        ms, r = timed(lambda: training.origin_train_step(m, mf, opt, ro, rd, gt, 64, 64, 0., 1., sigma_noise_std=0.1, as_float=False))
        assert np.isfinite(float(r["loss"]))
        return ms

    ms, ms_unfused = hip_iteration(True), hip_iteration(False)
    tm, tmf = _TorchNerf(synth.nerf_state(0)).cuda(), _TorchNerf(synth.nerf_state(1)).cuda()
    topt = torch.optim.Adam(list(tm.parameters()) + list(tmf.parameters()), lr=5e-4)
    ms_torch, loss_t = timed(lambda: _torch_origin_train_step(tm, tmf, topt, ro, rd, gt, 64, 64, 0.1))
    assert bool(torch.isfinite(loss_t))
    samples = rays * (64 + 128)
    flop = float(FLOP_TRAIN_PER_SAMPLE) * samples
    achieved = flop / (ms * 1e-3) / 1e12
    return {"metric": "ms per Origin_train iteration (1024 rays, 64 coarse + 128 fine-pass network samples, forward + backward + Adam) "
                      "on the fused HIP training kernels", "value": ms, "unit": "ms", "higher_is_better": False,
            "steps": steps, "dtype": "fp16x3", "network_samples_per_s": samples / (ms * 1e-3),
            "flop_per_iter": flop,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP16_TFLOPS,
                         "note": "algorithmic flops of forward + input gradients + weight gradients (%.3f MFLOP per network sample) / wall "
                                 "time of the whole iteration, sampling, compositing and Adam included; fp16x3 issues 3 MFMA products per "
                                 "algorithmic product; per-kernel times: profiles/r3_train_kernels.txt" % (FLOP_TRAIN_PER_SAMPLE / 1e6)},
            "comparator": {"what": "the same Origin_train iteration in stock torch-ROCm eager autograd (fp32 rocBLAS GEMMs, torch "
                                   "elementwise / cumprod / searchsorted kernels) on the same GPU, same rays; a baseline, not a target",
                           "value": ms_torch, "unit": "ms", "speedup": ms_torch / ms},
            "unfused": {"what": "the same iteration on the per-layer differentiable HIP dense layers (autograd_ops.py; round 2's path)",
                        "value": ms_unfused, "unit": "ms", "speedup": ms_unfused / ms}}


def make_renderer(precision, styled):
    from tgtc_style_amd import models, rendering, synth
    coarse, fine = build_nets(precision)
    style = None
    if styled:
        a = type("A", (NetArgs,), {"precision": precision.split("+")[-1]})
        cm, sm = models.StyleMLP_before_concat(a), models.StyleMLP_Wild_multilayers(a)
        cm.load_state_dict(t_state(synth.concat_state(2)))
        sm.load_state_dict(t_state(synth.style_state(3)))
        style = models.StylePair(cm.cuda(), sm.cuda())
    # (development: TGTC_BENCH_CHAIN forces the per-sample chain, TGTC_BENCH_SINGLE the single ray kernel where the library would split)
    fused = False if os.environ.get("TGTC_BENCH_CHAIN") else ("single" if os.environ.get("TGTC_BENCH_SINGLE") and not styled else True)
    return rendering.RayRenderer(coarse, fine, style, fused=fused)


def bench_frame(precision, h, w, steps, warmup, styled=False):
    """One GPU, one whole h x w frame per step (ray generation outside the events, the render inside)."""
    from tgtc_style_amd import synth, utils
    r = make_renderer(precision, styled)
    n = h * w
    z = torch.from_numpy(np.random.default_rng(4).standard_normal((n, 32)).astype(np.float32)).cuda() if styled else None
    focal = synth.fern_intrinsics(h, w)
    rays = [utils.gen_rays(h, w, focal, synth.spiral_pose(i % 120)) for i in range(2)]
    dt, ms, out = time_loop(lambda i: r.render(*rays[i % 2], N_COARSE, N_FINE, near=0., far=1., z=z), steps, warmup)
    assert bool(torch.isfinite(out["rgb"]).all()) and bool(torch.isfinite(out["t"]).all())
    flop = (FLOP_PER_RAY_STYLED if styled else FLOP_PER_RAY) * n
    roof = {"bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
            "frac": flop / (ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, "kernel_ms": ms, "algorithmic_tflop_per_launch": flop / 1e12}
    if styled and (h, w) == (H, W):     # the PMC passes of tools/profile_r*.sh profiled whole 400x400 stylised frames
        traffic, provenance = pmc_traffic("styled:fp16x3+fp16mx")
        algo = n * BYTES_PER_RAY_STYLED
        roof.update(traffic=traffic, traffic_source=provenance, algorithmic_bytes_per_launch=algo,
                    traffic_over_algorithmic=(traffic / algo) if traffic else None)
    return {"value": n * steps / dt, "unit": "rays/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": precision,
            "rays_per_step": n, "kernel_ms": ms, "roofline": roof}


def run_headline(args, precision, rank, world, dist):
    """The timed headline run; returns the JSON dict on rank 0, None elsewhere."""
    from tgtc_style_amd import parallel, synth, utils
    renderer = make_renderer(precision, False)
    split = renderer.fused is True and renderer._split_is_faster()     # the library's choice for fp16x3 + fp16mx (include/tgtc_hip.h)
    fused = renderer.fused and renderer._fused_shape(N_COARSE, N_FINE) and not split
    focal = synth.fern_intrinsics(H, W)
    n_rays = H * W
    by_rays = args.sharding == "rays" and world > 1
    lo, hi = parallel.shard_range(n_rays, rank, world) if by_rays else (0, n_rays)
    images = [torch.empty(hi - lo, 4, device="cuda", dtype=torch.float32) for _ in range(2)]
    timed = Timed(args.steps)
    # N > 1: the gather of frame i runs on a side stream while frame i+1 renders; two image buffers
    gather = None
    if world > 1:
        gather = parallel.AsyncGather((lambda x: parallel.gather_rows(x, n_rays, rank, world, dist)) if by_rays
                                      else (lambda x: parallel.gather_frames(x, world, dist)))

    def step(i, timed_idx=None):
        # frames sharding: every rank its own pose; rays sharding: all ranks the same pose, each its own pixel range
        pose = synth.spiral_pose((i if by_rays else i * world + rank) % 120)
        o, d = utils.gen_rays(H, W, focal, pose, first_pixel=lo, n=hi - lo)
        render = lambda: renderer.render(o, d, N_COARSE, N_FINE, near=0., far=1.)
        out = timed.run(timed_idx, render) if timed_idx is not None else render()
        image = images[i % 2]
        if gather is not None:
            gather.reusable(i)
        image[:, :3] = out["rgb"]
        image[:, 3] = out["t"]
        if gather is None:
            return image
        gather.submit(image)
        return None

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame = step(args.warmup + i, timed_idx=i)
    if gather is not None:
        frame = gather.result()        # the last frame's gather is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    assert bool(torch.isfinite(frame).all())
    if rank != 0:
        return None

    render_ms = timed.mean_ms()
    rays_total = (1 if by_rays else world) * n_rays * args.steps
    rays_launch = hi - lo
    pf = precision.split("+")[-1]
    whole = {"ms": render_ms, "achieved": float(FLOP_PER_RAY) * rays_launch / (render_ms * 1e-3) / 1e12,
             "mfma_products_per_algorithmic_product": mfma_per_product(precision)}
    whole["frac"] = whole["achieved"] / PEAK_FP16_TFLOPS
    whole["mfma_pipe_frac"] = whole["frac"] * whole["mfma_products_per_algorithmic_product"]
    if split:
        # the dominant kernel of the split path: the fine pass (192 depths per ray, full network) on the two-tile per-sample
        # kernel -- ONE launch per frame, timed live here on the launch stream with the frame's ray count
        from tgtc_style_amd import hip
        o, d = utils.gen_rays(H, W, focal, synth.spiral_pose(0), first_pixel=lo, n=hi - lo)
        n_all = N_COARSE + N_FINE
        ts = torch.linspace(0., 1., n_all, device="cuda").expand(rays_launch, n_all).contiguous()
        rgb_s, sig_s = torch.empty(rays_launch * n_all, 3, device="cuda"), torch.empty(rays_launch * n_all, device="cuda")
        lib, handle = hip.load(), renderer.fine.packed().handle
        fine_pass = lambda i: hip.check(lib.tgtc_nerf_forward_rays(handle, hip.ptr(o), hip.ptr(d), hip.ptr(ts), rays_launch, n_all,
                                                                   hip.ptr(rgb_s), hip.ptr(sig_s), hip.stream()))
        _, kernel_ms, _ = time_loop(fine_pass, max(args.steps, 3), 1)
        flop_launch = 2.0 * MAC_FULL * rays_launch * n_all
        # the frame's second kernel, timed the same way: the coarse pass (fp16x3, densities only) on its two-tile kernel
        ts_c = torch.linspace(0., 1., N_COARSE, device="cuda").expand(rays_launch, N_COARSE).contiguous()
        sig_c = torch.empty(rays_launch * N_COARSE, device="cuda")
        h_c = renderer.coarse.packed().handle
        coarse_pass = lambda i: hip.check(lib.tgtc_nerf_forward_rays(h_c, hip.ptr(o), hip.ptr(d), hip.ptr(ts_c), rays_launch, N_COARSE,
                                                                     None, hip.ptr(sig_c), hip.stream()))
        _, coarse_ms, _ = time_loop(coarse_pass, max(args.steps, 3), 1)
        coarse_flop = 2.0 * MAC_SIGMA * rays_launch * N_COARSE
        whole["kernels"] = {
            "nerf_mx2_kernel (fine pass, fp16mx)": {"ms": kernel_ms, "frac": flop_launch / (kernel_ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS,
                                                    "mfma_pipe_frac": flop_launch * MFMA_PER_PRODUCT["fp16mx"] / (kernel_ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS},
            "nerf_x3s_kernel (coarse pass, fp16x3, densities)": {"ms": coarse_ms, "frac": coarse_flop / (coarse_ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS,
                                                                 "mfma_pipe_frac": coarse_flop * MFMA_PER_PRODUCT["fp16x3"] / (coarse_ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS}}
        kname = ("nerf_mx2_kernel (the fine pass of %d rays in one launch: PE + full NeRF MLP on %d depths per ray; the frame's other "
                 "launches: ray generation, coarse depths, coarse MLP (fp16x3, sigma only), two composites, fine sampling -- "
                 "`whole_render` is all of them)" % (rays_launch, n_all))
        traffic, provenance = pmc_traffic("nerf_mx2_kernel")
        algo_bytes = rays_launch * (2 * 3 * 8 + n_all * 4 + n_all * 16)      # rays + depths in, rgb + sigma per sample out
        prod = MFMA_PER_PRODUCT["fp16mx"]
    else:
        kernel_ms = render_ms
        flop_launch = float(FLOP_PER_RAY) * rays_launch
        kname = ("%s (whole render of %d rays in one launch: PE + coarse MLP x128, fine sampling, PE + fine MLP x192, compositing)"
                 % ("fused_render_kernel" if fused else "per-sample kernel chain", rays_launch))
        traffic, provenance = pmc_traffic("fused_render_kernel:%s" % precision) if fused else (None, "chain: not profiled")
        algo_bytes = rays_launch * BYTES_PER_RAY
        prod = mfma_per_product(precision)
    achieved = flop_launch / (kernel_ms * 1e-3) / 1e12
    if traffic and rays_launch != n_rays:   # the PMC pass profiled whole-frame launches; a ray range streams in proportion
        traffic *= rays_launch / n_rays
        provenance += "; scaled by %d / %d rays of this launch" % (rays_launch, n_rays)
    return {
        "metric": "rays/sec (128c+64f samples) on fern 400x400",
        "value": rays_total / dt,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if by_rays else "weak",
        "vs_baseline": None,
        "dtype": DTYPE[pf] + ("" if "+" not in precision else " (fine pass; coarse pass: %s)" % precision.split("+")[0]),
        "data": "synthetic",
        "config": {"workload": "fern-shaped 400x400 frame, plain NeRF render (style off), 128 coarse + 64 fine samples/ray, "
                               + ("one frame per step, rays sharded over the ranks" if by_rays else "one whole frame per rank per step")
                               + ", seeded random-init weights",
                   "rays_per_step": (1 if by_rays else world) * n_rays, "precision": precision,
                   "sharding": "rays" if by_rays else "frames", "algorithmic_mflop_per_ray": FLOP_PER_RAY / 1e6,
                   "precision_note": NOTES.get(precision, "")},
        "roofline": {"bound": "mfma", "kernel": kname,
                     "achieved": achieved, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP16_TFLOPS,
                     "traffic": traffic, "traffic_source": provenance,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None,
                     "kernel_ms": kernel_ms, "algorithmic_tflop_per_launch": flop_launch / 1e12,
                     "mfma_products_per_algorithmic_product": prod,
                     "mfma_pipe_frac": achieved * prod / PEAK_FP16_TFLOPS},
        # every launch of one frame's render together (what `value` is made of), priced like rounds 1-3 priced the fused kernel
        "whole_render": whole,
        "render_path": "split (per-sample kernels; fine pass on the two-tile kernel)" if split else ("fused ray kernel" if fused else "per-sample kernel chain"),
        "whole_path_tflops": rays_total / dt * FLOP_PER_RAY / 1e12,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="fp16x3+fp16mx",
                    help="fp16x3 | fp16mx | fp16, or coarse+fine (default: fp16x3+fp16mx, the fastest mode that passes every 1e-3 parity test)")
    ap.add_argument("--sharding", default="frames", choices=["frames", "rays"],
                    help="N > 1: frames = every rank a whole frame per step (weak scaling, BASELINE config 5 style); "
                         "rays = one frame per step, contiguous ray ranges per rank (BASELINE config 4, strong scaling)")
    ap.add_argument("--cpu-rays", type=int, default=8192, help="rays of the CPU baseline sample (0 disables)")
    ap.add_argument("--alt-precision", default="fp16x3,fp16",
                    help="further precisions (comma separated) reported under `alt_precisions` ('' disables)")
    ap.add_argument("--configs", default="styled,style2d,trex_rays,train_step",
                    help="N = 1: further BASELINE configs measured after the headline and reported under `configs` ('' disables)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # one rank per GPU; TGTC_DIST_BACKEND=gloo lets several ranks share one GPU for functional rehearsals of the N>1 path
    backend = os.environ.get("TGTC_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    dist = None
    if world > 1:   # the process group comes up before the first HIP call of this process
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)

    line = run_headline(args, args.precision, rank, world, dist)
    alts = [p for p in args.alt_precision.split(",") if p and p != args.precision]
    alt_lines = [run_headline(args, p, rank, world, dist) for p in alts]   # every rank joins the collectives
    if rank == 0:
        for p, alt in zip(alts, alt_lines):
            entry = {k: alt[k] for k in ("value", "ms_per_step", "dtype")}
            entry.update(precision=p, roofline_frac=alt["roofline"]["frac"], kernel_ms=alt["roofline"]["kernel_ms"],
                         mfma_pipe_frac=alt["roofline"]["mfma_pipe_frac"], note=NOTES.get(p, ""))
            line.setdefault("alt_precisions", []).append(entry)
        if world == 1:
            short = int(os.environ.get("TGTC_BENCH_CONFIG_STEPS", max(2, min(args.steps, 3))))   # (profiling runs vary it)
            cfg = {}
            wanted = [c for c in args.configs.split(",") if c]
            if "styled" in wanted:      # BASELINE config 3, ray path: the stylised chain (NeRF + concat MLP + style MLP)
                cfg["styled"] = dict(bench_frame("fp16x3", H, W, short, 1, styled=True),
                                     metric="rays/sec (128c+64f) on fern 400x400, stylised (concat + style MLPs)")
            if "style2d" in wanted:     # BASELINE config 3, per-frame ViT + CNN decoder + VGG pass
                cfg["style2d"] = bench_style2d("fp16x3", short, 1)
            if "trex_rays" in wanted:   # BASELINE config 4's frame on one GPU (its 8 ray ranges are this frame's slices)
                cfg["trex_rays"] = dict(bench_frame(args.precision, 378, 504, short, 1),
                                        metric="rays/sec (128c+64f) on a whole trex 504x378 frame (190512 rays), one GPU")
            if "train_step" in wanted:  # SURVEY 8f rank 4 (training side)
                cfg["train_step"] = bench_train_step(6, 3)
            if cfg:
                line["configs"] = cfg
            if args.cpu_rays > 0:
                line["cpu_baseline"] = cpu_baseline(args.cpu_rays)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
