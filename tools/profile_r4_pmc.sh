#!/bin/bash
# Round-4 counter passes on the HEADLINE ALONE (the counters that name the stall): instruction mix, wait
# classes, LDS.  Counter passes carry no trace domain; the program follows `--` directly.  Summary: tools/pmc_summary.py.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TOP=$R/gpurun_out/prof_r4
mkdir -p $TOP
cd $R
rocprofv3 -L > $TOP/counters_list.txt 2>&1 || true
ARGS="--steps 4 --warmup 2 --cpu-rays 0 --alt-precision= --configs= --precision ${PREC:-fp16x3+fp16mx}"
OUT=$TOP/${TAG:-headline_x3mx}; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d $OUT/sq -- python3 bench.py $ARGS > $OUT/sq.json 2> $OUT/sq.err
echo "pass sq exit $?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM \
    --output-format csv -d $OUT/sq2 -- python3 bench.py $ARGS > $OUT/sq2.json 2> $OUT/sq2.err
echo "pass sq2 exit $?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq3 -- python3 bench.py $ARGS > $OUT/sq3.json 2> $OUT/sq3.err
echo "pass sq3 exit $?"
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "sq*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(root, "summary.txt"), "w") as f:
    for k, cs in acc.items():
        if "fused_render" not in k and "nerf" not in k and "styled" not in k:
            continue
        f.write(k + "\n")
        for c, v in sorted(cs.items()):
            f.write("   %-34s n=%-3d mean %.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(os.path.join(root, "summary.txt")).read())
PY
