#!/usr/bin/env python3
"""Copy the summaries of tools/profile_r3.sh (gpurun_out/prof_r3/<run>/) into profiles/ and rewrite profiles/pmc_traffic.json.

Every entry names its dispatch mix (one workload per profiled run) and carries the raw FETCH_SIZE figure next to the
corrected one: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced stream of 16 bytes per lane
(`global_load` and `buffer_load ... lds` alike; MI355X_MICROARCH.md section HBM), which is the access pattern of the LDS-DMA
weight streams -- all but ~8 MB of what the fused ray kernel fetches -- and of the 16-byte-per-lane slab reloads of the
stylised kernel.  WRITE_SIZE is exact for 16-byte-per-lane stores and is taken as read."""
import csv, glob, json, os, shutil, subprocess, sys
top = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_r4'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r4'


def mean(run, sub, counter, key):
    v = []
    for p in glob.glob('%s/%s/%s/**/*counter_collection.csv' % (top, run, sub), recursive=True):
        for r in csv.DictReader(open(p)):
            if r['Counter_Name'] == counter and key in r['Kernel_Name']:
                v.append(float(r['Counter_Value']))
    return (sum(v) / len(v), len(v)) if v else (None, 0)


head = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD']).decode().strip()
d = {"collected": "tools/profile_%s.sh on MI355X" % tag + ": rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of "
                  "`python3 bench.py --steps 4 --warmup 2 --cpu-rays 0 --alt-precision= --configs=<one> --precision <p>`; means per dispatch; "
                  "summaries profiles/%s_<run>_rocprof_summary.txt" % tag,
     "commit": head,
     "note": "FETCH_SIZE / WRITE_SIZE are in KB and count L2<->fabric requests (Infinity-Cache hits included).  fetch_bytes = 2 x "
             "fetch_bytes_raw: the guide's gfx950 correction for 16-byte-per-lane coalesced streams (LDS-DMA weight streams, slab "
             "reloads).  The fetches of the fused ray kernel are its two weight streams being re-served from the Infinity Cache into "
             "an XCD's 4 MiB L2 at each coarse<->fine phase change; rays in and pixels out are 10.2 MB.",
     "kernels": {}}
for name, run, key, what in (
        ("nerf_mx2_kernel", "headline_x3mx", "nerf_mx2_kernel", "nerf_mx2_kernel<IN_RAYS>, the fine pass of 160000 rays x 192 depths per dispatch (headline alone, split path)"),
        ("nerf_x3s_kernel", "headline_x3mx", "nerf_x3s_kernel", "nerf_x3s_kernel, the coarse pass of 160000 rays x 128 depths per dispatch (headline alone, split path)"),
        ("fused_render_kernel:fp16x3+fp16mx", "headline_x3mx_single", "fused_render_kernel<0, 2>", "fused_render_kernel<0,2>, 160000 rays of a 400x400 frame per dispatch (TGTC_BENCH_SINGLE=1)"),
        ("fused_render_kernel:fp16x3", "headline_x3", "fused_render_kernel<0, 0>", "fused_render_kernel<0,0>, 160000 rays of a 400x400 frame per dispatch (headline alone)"),
        ("styled:fp16x3+fp16mx", "styled", "styled", "stylised render kernel(s), 160000 rays of a 400x400 frame per dispatch (--configs styled)")):
    f, n = mean(run, 'fetch', 'FETCH_SIZE', key)
    w, _ = mean(run, 'write', 'WRITE_SIZE', key)
    if f is None or w is None:
        print("no counters for", name)
        continue
    # the per-sample kernels fetch their depths with 4-byte-per-lane loads (FETCH_SIZE exact: calibrated in round 1 on this very
    # access, DESIGN.md section 3.1 Roofline) and 2 MB of weights per XCD; the x2 of the guide applies to 16-byte-per-lane streams
    k = 1.0 if name.startswith("nerf_") else 2.0
    d['kernels'][name] = {"fetch_bytes_raw": f * 1024, "fetch_bytes": k * f * 1024, "fetch_correction": k, "write_bytes": w * 1024,
                          "source": "%s, n=%d dispatches" % (what, n)}
    print(name, round(k * f * 1024 / 1e9, 3), 'GB fetched (corrected x%g)' % k, round(w * 1024 / 1e9, 4), 'GB written', n)
json.dump(d, open('profiles/pmc_traffic.json', 'w'), indent=1)
for run in ("headline_x3mx", "headline_x3mx_single", "headline_x3", "styled"):
    if os.path.exists('%s/%s/summary.txt' % (top, run)):
        shutil.copy('%s/%s/summary.txt' % (top, run), 'profiles/%s_%s_rocprof_summary.txt' % (tag, run))
        shutil.copy('%s/%s/bench_stats.json' % (top, run), 'profiles/%s_%s_bench_under_rocprof.json' % (tag, run))
        stats = sorted(glob.glob('%s/%s/stats/**/*kernel_stats.csv' % (top, run), recursive=True), key=os.path.getmtime)
        if stats:
            shutil.copy(stats[-1], 'profiles/%s_%s_kernel_stats.csv' % (tag, run))
