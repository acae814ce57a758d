#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/profile_r3.sh: per-kernel mean duration, FETCH_SIZE / WRITE_SIZE per dispatch,
SQ counters per dispatch.  Kernel names are shortened to their function name + template head."""
import csv, glob, os, re, sys, collections
root = sys.argv[1]

def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("tgtc::", "")
    return name[:110]

def rows(pattern):
    for path in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r

# 1. kernel stats
stats = [r for r in rows("stats/**/*kernel_stats.csv")]
print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for r in sorted(stats, key=lambda r: -float(r.get("TotalDurationNs", r.get("Total_Duration(ns)", 0)) or 0))[:14]:
    name = r.get("Name", r.get("KernelName", "?"))
    calls = r.get("Calls", "?")
    avg = float(r.get("AverageNs", r.get("Average(ns)", 0)) or 0)
    pct = r.get("Percentage", r.get("Percentage(%)", "?"))
    print("%9.3f ms avg  x%-4s %6s%%  %s" % (avg / 1e6, calls, pct, short(name)))

# 2/3. counters
for sub in ("fetch", "write", "sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(sub + "/**/*counter_collection.csv"):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== counters:", sub)
    for k, cs in sorted(acc.items(), key=lambda kv: -max(sum(v) for v in kv[1].values())):
        if not any(x in k for x in ("fused_render", "styled", "nerf_mlp", "nerf_mx", "nerf_x3s", "gemm_kernel", "attn")):
            continue
        print(" ", k)
        for c, v in sorted(cs.items()):
            print("     %-28s n=%-3d mean %.6g" % (c, len(v), sum(v) / len(v)))
