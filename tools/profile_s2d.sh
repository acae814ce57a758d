#!/bin/bash
# Per-kernel times of ONE steady-state frame of the 2-D style pass (bench.py --configs style2d only).  Two profiled runs with
# different step counts; their difference, divided by the extra frames, is what a frame costs -- handle creation (one
# host-to-device upload per weight tensor: ~300 `__amd_rocclr_copyBuffer` blits) and warm-up drop out.  (Round 2's table
# divided the whole process by the frame count, set-up copies included, which VERDICT r3 read as 312 copies per frame.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_s2d; rm -rf $OUT; mkdir -p $OUT; cd $R
for S in 4 12; do
  TGTC_BENCH_CONFIG_STEPS=$S rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$S -- python3 bench.py --steps 2 --warmup 2 --cpu-rays 0 --alt-precision "" --configs style2d > $OUT/bench_$S.json 2> $OUT/err_$S.txt
done
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_s2d"
def load(s):
    rows = {}
    for p in glob.glob(root + "/stats_%d/**/*kernel_stats.csv" % s, recursive=True):
        for r in csv.DictReader(open(p)):
            rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return rows
a, b = load(4), load(12)
frames = lambda rows: sum(c for n, (c, t) in rows.items() if "style_feature_kernel" in n)
df = frames(b) - frames(a)
out = []
for n in b:
    if "fused_render" in n:
        continue
    c = b[n][0] - a.get(n, (0, 0))[0]
    t = b[n][1] - a.get(n, (0, 0))[1]
    out.append((t / df / 1e6, c / df, n))
out.sort(reverse=True)
lines = ["per steady-state frame = (run with %d stylised frames - run with %d) / %d" % (frames(b), frames(a), df)]
for ms, c, n in out[:20]:
    lines.append("%8.3f ms/frame  %6.1f launches/frame  %s" % (ms, c, n[:120]))
lines.append("total %.2f ms and %.0f launches per frame" % (sum(o[0] for o in out), sum(o[1] for o in out)))
lines.append("copyBuffer launches per steady-state frame: %.2f (whole process: %d at 4 steps, %d at 12)" % (
    sum(o[1] for o in out if "copyBuffer" in o[2] and "Rect" not in o[2]),
    sum(c for n, (c, t) in a.items() if "copyBuffer" in n and "Rect" not in n), sum(c for n, (c, t) in b.items() if "copyBuffer" in n and "Rect" not in n)))
open(root + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
