#!/bin/bash
# per-kernel times of the 2-D style pass (bench.py --configs style2d only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_s2d; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 6 --warmup 2 --cpu-rays 0 --alt-precision "" --configs style2d > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv,glob,os
root=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_s2d"
for p in glob.glob(root+"/stats/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(p)))
    rows=[r for r in rows if "fused_render" not in r["Name"]]
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    frames=max(1,sum(int(r["Calls"]) for r in rows if "style_feature_kernel" in r["Name"]))   # one call per stylised frame
    for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:16]:
        print("%8.3f ms/frame  avg %8.1f us x%-5s %s" % (float(r["TotalDurationNs"])/1e6/frames, float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:120]))
    print("total %.2f ms per frame over %d frames (all kernels of the process except the ray kernel, set-up copies included)" % (tot/1e6/frames, frames))
PY
