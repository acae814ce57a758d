#!/bin/bash
# instruction-cache counters of the fused ray kernel (its fully unrolled layer chains are ~200 KB of code)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_icache; mkdir -p $OUT; cd $R
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_[A-Z_]*" $OUT/avail.txt | sort -u | tr '\n' ' '; echo
ARGS="--steps 3 --warmup 1 --cpu-rays 0 --alt-precision  --configs "
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 bench.py --steps 3 --warmup 1 --cpu-rays 0 --alt-precision "" --configs "" > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES --output-format csv -d $OUT/b -- python3 bench.py --steps 3 --warmup 1 --cpu-rays 0 --alt-precision "" --configs "" > /dev/null 2> $OUT/b.err
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_icache"
for sub in "ab":
    acc=collections.defaultdict(list)
    for p in glob.glob(root+"/"+sub+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "fused_render" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(sub, k, len(v), "%.4g" % (sum(v)/len(v)))
    if not acc: print(sub, "no counters;", open(root+"/"+sub+".err").read()[-600:])
PY
