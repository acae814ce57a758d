"""Test helper: the parity statement for rendered rays (frozen in round 4; rounds 2-3 moved it, see the end of this text).

The reference's render chain is discontinuous and, next to its discontinuities, ill-conditioned: `sample_pdf` interpolates a
fine sample's depth as  b_lo + (u - cdf_lo) / den * (b_hi - b_lo)  with  den = cdf_hi - cdf_lo  (utils.py:601-607), so the
depth follows the coarse weights with a gain of (bin width) / den -- 1e3 ... 1e5 in empty space -- and den < 1e-5 switches
the formula (utils.py:604-605).  Two float32 evaluations of the REFERENCE that differ only in summation order (another BLAS,
another vector width in `torch.sum` / `cumsum`) differ by ~1e-6 in the weights and therefore by up to a bin in such a
sample; on the seeded scenes 0.5-4 % of the rays move by more than 1e-4 between the float32 and the float64 oracle.  One
reference output cannot be the yardstick on those rays.  The statement a render has to meet, for EVERY ray:

  (1)  its output lies within `tol` of ONE of the admissible outputs of a FIXED variant set, evaluated for all rays alike:
       the oracle in float32 (the reference's arithmetic), in float64, with either half of the chain in each precision, and
       with the ray origin moved by +-1e-7 relative;                                                 ["explained"]
  or
  (2)  it carries the STAGE CERTIFICATE -- three deterministic checks against the oracle run on the HIP path's own
       intermediate values (no sampling, no tolerance that grows with the miss):
         a. coarse stage: the HIP coarse weights equal the float32 oracle's to 2e-5 (measured 2e-6: fp32 level);
         b. sampler: every merged depth lies in what sample_pdf can return on the SAME (HIP) coarse weights when its cdf is
            perturbed by at most 64 float32 ulps (4e-6: the worst-case rounding of a 62-term float32 sum): the interpolated
            depth within  bin * min(1, 2 * 4e-6 / den),  anywhere in the bin where den is within a factor two of the 1e-5
            switch, the neighbouring bin where u sits within 4e-6 of a cdf edge (sampler_bound);
         c. fine stage (conditional parity): the float32 oracle's fine network + compositing evaluated AT THE HIP DEPTHS
            reproduces the HIP output to `tol`.
       A kernel error in any stage fails its check; what a certified ray is allowed is only the sampler's branch.
       The share of certified rays is printed and bounded (`max_certified`).

On top of that two statistics are asserted and printed: at least 95 % of the rays lie within `tol` of the float32 oracle
itself (strict reading, no variants), and the share of rays whose variant outputs disagree by more than 1e-4
("ill-conditioned in the reference") stays below `max_ill`.  The 99th percentile of the distance to the float32 oracle is
printed.

History: round 2 exempted ill-conditioned rays; round 3 compared with the per-channel RANGE of the variants and, for rays
that missed, drew further random perturbations (32 x 1e-7, then 48 with 1e-6) -- an acceptance set that grew where the kernel
disagreed.  Round 4 dissected the ray that had made round 3 widen it (flower 64c+64f, ray 498 of the 40x40 frame,
tests/probes/analyse_llff_config.py): the float32 and float64 oracles differ by 1.5e-2 on it, one fine sample is interpolated
with den = 2.8e-5, coarse weights that agree to 2e-6 move it by 7.6e-4, and the oracle's fine pass on the HIP depths
reproduces the HIP pixel to 2e-7 (fp16x3) / 9e-6 (fp16mx): the sampler's conditioning, not the fp16mx arithmetic."""
import torch

from oracle import raymarch


def _vec(out):
    return torch.cat([out["rgb_fine"].float(), out["t_fine"].float()[:, None]], 1)       # [R, 4]


def hip_stages(coarse, ro, rd, n_coarse, n_fine, near=0., far=1., jitter=None):
    """The HIP path's own intermediates for the stage certificate, through the per-stage operators (the fused ray kernel runs
    the same per-tile code; tests/test_fused_gpu.py holds the two paths together): coarse depths, coarse weights, merged fine
    depths.  `coarse` = the coarse models.StyleNerf; ro, rd on the GPU."""
    from tgtc_style_amd import utils
    pts, ts = utils.sampling_pts_uniform(ro, rd, n_coarse, near=near, far=far, jitter=jitter)
    c = coarse(pts=pts, dirs=rd[:, None, :])
    _, _, w = utils.alpha_composition(c["rgb"], c["sigma"], ts)
    _, ts_f = utils.sampling_pts_fine_torch(ro, rd, ts, w, n_fine)
    return {"ts_c": ts.cpu(), "w_c": w.cpu(), "ts_f": ts_f.cpu()}


def sampler_bound(ts_c, w_c, ts_f, n_fine, delta=4e-6):
    """-> excess [r]: by how much the worst merged depth of `ts_f` lies outside what sample_pdf (utils.py:583-609, det) may
    return on (ts_c, w_c) when its cdf is perturbed by at most `delta` (64 float32 ulps); 0 = inside.  For fine sample j
    (u_j = j / (n_fine - 1)) every bin k whose perturbed cdf interval can contain u_j is a candidate; in a candidate bin the
    depth is mid_k + clamp((u - cdf_k) / den_k) * width_k within width_k * min(1, 2 delta / den_k), anywhere in the bin
    where den_k is within a factor two of the 1e-5 switch of utils.py:604-605, and mid[-1] when u_j falls behind the
    last cdf value (which is 1 only up to rounding, utils.py:595).  The coarse depths must appear unchanged.  float64, a loop
    per ray: it runs on the handful of rays that need the certificate."""
    import numpy as np
    out = []
    for tc, wc, tf in zip(ts_c.double().numpy(), w_c.double().numpy(), ts_f.double().numpy()):
        mid = 0.5 * (tc[1:] + tc[:-1])
        w = wc[1:-1] + 1e-5
        cdf = np.concatenate([[0.0], np.cumsum(w / w.sum())])
        lo_t, hi_t = np.empty(n_fine), np.empty(n_fine)
        for j, u in enumerate(np.linspace(0.0, 1.0, n_fine)):
            a, b = np.inf, -np.inf
            for k in range(len(cdf) - 1):
                if cdf[k] - delta <= u <= cdf[k + 1] + delta:
                    den, width = cdf[k + 1] - cdf[k], mid[k + 1] - mid[k]
                    if den < 2e-5:
                        a, b = min(a, mid[k]), max(b, mid[k + 1])
                    else:
                        t0 = mid[k] + min(max((u - cdf[k]) / den, 0.0), 1.0) * width
                        r = width * min(1.0, 2 * delta / den)
                        a, b = min(a, max(t0 - r, mid[k])), max(b, min(t0 + r, mid[k + 1]))
            if u >= cdf[-1] - delta:            # searchsorted past the end: below = above = last bin edge
                a, b = min(a, mid[-1]), max(b, mid[-1])
            if u <= cdf[0] + delta:
                a, b = min(a, mid[0]), max(b, mid[0])
            lo_t[j], hi_t[j] = a, b
        # the coarse depths must be in the merged list as they are; what remains are the fine samples, ascending
        rest, i, worst = [], 0, 0.0
        for v in tf:
            if i < len(tc) and abs(v - tc[i]) <= 1e-7:
                i += 1
            else:
                rest.append(v)
        if i != len(tc) or len(rest) != n_fine:
            out.append(1.0)
            continue
        lo_s, hi_s = np.sort(lo_t), np.sort(hi_t)
        for v, a, b in zip(rest, lo_s, hi_s):
            worst = max(worst, a - 1e-6 - v, v - b - 1e-6)
        out.append(max(worst, 0.0))
    return torch.tensor(out, dtype=torch.float64)


def check(label, rgb, t, render, ro, rd, tol=1e-3, max_ill=0.12, median_bound=None, mixed=True, stages=None, n_fine=None,
          max_certified=0.02, min_strict=0.95):
    """render(rays_o, rays_d, dtype_coarse, dtype_fine, sel, ts_fine=None) -> oracle dict with rgb_fine / t_fine / w_coarse
    (CPU tensors); sel is None for all rays or the index tensor of the subset the rays were taken from (per-ray side inputs
    such as latents); ts_fine, when given, replaces the oracle's own fine depths (check 2c).
    stages(sel) -> dict ts_c, w_c, ts_f of the HIP path for the rays `sel` (hip_stages); needed only if a ray is not explained
    by the fixed variant set.  Asserts the statement above; returns (per-ray distance to the nearest admissible output,
    ill-conditioned mask)."""
    f32, f64 = torch.float32, torch.float64
    x = torch.cat([rgb.detach().cpu().float(), t.detach().cpu().float()[:, None]], 1)
    base_out = render(ro, rd, f32, f32, None)
    vs = [_vec(base_out), _vec(render(ro, rd, f64, f64, None))]
    if mixed:
        vs += [_vec(render(ro, rd, f64, f32, None)), _vec(render(ro, rd, f32, f64, None))]
    vs += [_vec(render(ro * (1.0 + 1e-7), rd, f32, f32, None)), _vec(render(ro * (1.0 - 1e-7), rd, f32, f32, None))]
    V = torch.stack(vs)                                                     # [variants, R, 4]
    e0 = (x - V[0]).abs().max(1).values                                     # against the float32 oracle alone
    near = (x[None] - V).abs().max(2).values.min(0).values                  # nearest single admissible output
    ill = (V.max(0).values - V.min(0).values).max(1).values > 1e-4
    miss = torch.nonzero(near > tol).flatten()
    certified = torch.zeros_like(near, dtype=torch.bool)
    notes = []
    if miss.numel():
        assert stages is not None and n_fine is not None, (
            "%s: %d rays are farther than %g from every output of the fixed variant set (worst %.3e) and the test provides no "
            "stage certificate" % (label, miss.numel(), tol, float(near.max())))
        st = stages(miss)
        o, d = ro[miss], rd[miss]
        w_or = base_out["w_coarse"][miss].float()
        a = (st["w_c"].float() - w_or).abs().max(1).values                                       # 2a
        b = sampler_bound(st["ts_c"], st["w_c"], st["ts_f"], n_fine)                             # 2b
        cond = _vec(render(o, d, f32, f32, miss, ts_fine=st["ts_f"]))
        c = (x[miss] - cond).abs().max(1).values                                                 # 2c
        ok = (a <= 2e-5) & (b <= 0) & (c <= tol)
        certified[miss] = ok
        for i, r in enumerate(miss.tolist()):
            notes.append("ray %d: nearest variant %.2e; coarse weights %.1e, sampler excess %.1e, conditional parity %.1e -> %s" % (
                r, float(near[r]), float(a[i]), float(b[i]), float(c[i]), "certified" if bool(ok[i]) else "FAILED"))
    passed = (near <= tol) | certified
    well = ~ill
    p99 = float(torch.quantile(e0, 0.99)) if e0.numel() > 1 else float(e0.max())
    print("%s, %d rays: nearest admissible output max %.2e on the %d rays explained by the fixed variant set; %d rays (%.2f %%) "
          "carry the stage certificate; float32 oracle alone: %.1f %% within %g, p99 %.2e, max %.2e on the %d well-conditioned rays, "
          "median %.2e; %d rays (%.1f %%) ill-conditioned in the reference itself" % (
              label, near.numel(), float(near[near <= tol].max()) if bool((near <= tol).any()) else 0.0, int((near <= tol).sum()),
              int(certified.sum()), 100.0 * float(certified.float().mean()), 100.0 * float((e0 <= tol).float().mean()), tol, p99,
              float(e0[well].max()) if bool(well.any()) else 0.0, int(well.sum()), float(e0.median()), int(ill.sum()),
              100.0 * float(ill.float().mean())))
    for n in notes:
        print("   " + n)
    assert bool(passed.all()), "%s: %d rays neither within %g of an admissible output nor certified:\n%s" % (
        label, int((~passed).sum()), tol, "\n".join(notes))
    assert float(certified.float().mean()) <= max_certified, "%s: %.2f %% of the rays needed the stage certificate" % (
        label, 100.0 * float(certified.float().mean()))
    assert float((e0 <= tol).float().mean()) >= min_strict, "%s: only %.1f %% of the rays within %g of the float32 oracle" % (
        label, 100.0 * float((e0 <= tol).float().mean()), tol)
    assert float(ill.float().mean()) <= max_ill, "%s: %.1f %% of the rays ill-conditioned" % (label, 100.0 * float(ill.float().mean()))
    if median_bound is not None:
        assert float(e0.median()) <= median_bound
    return torch.where(certified, torch.zeros_like(near), near), ill
