cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_st; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/time_style_train.py 1024 > $OUT/out.txt 2> $OUT/err.txt
tail -1 $OUT/out.txt
python3 - <<'PY'
import csv,glob,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_st"
p=sorted(glob.glob(root+"/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(p)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    print("%7.3f ms/iter avg %7.1f us x%-5s %s"%(float(r["TotalDurationNs"])/1e6/13,float(r["AverageNs"])/1e3,r["Calls"],r["Name"][:100]))
print("kernel time %.2f ms per iteration"%(tot/1e6/13))
PY
