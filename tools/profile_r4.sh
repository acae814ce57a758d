#!/bin/bash
# Round-4 profiles: every precision of the HEADLINE ALONE, one workload per run (round 2 mixed the headline with the configs);
# fp16x3+fp16mx twice -- the split path the library picks and, TGTC_BENCH_SINGLE=1, the single ray kernel --
# then strict fp16x3 and the stylised config by itself.
#   per run:  1. rocprofv3 --kernel-trace --stats      2./3. --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)
#             4./5. --pmc SQ_* GRBM_GUI_ACTIVE: wait classes, then instruction mix + LDS (VERDICT r3 item 1's list; counter passes
#                   carry no trace domain; the program follows `--` directly)
# Summaries: tools/pmc_summary.py <dir>; tools/refresh_profiles.py copies them into profiles/ and rewrites pmc_traffic.json.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TOP=$R/gpurun_out/prof_r4
rm -rf $TOP; mkdir -p $TOP
cd $R
run() {   # run <tag> <bench args...>
  local OUT=$TOP/$1; shift; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py "$@" > $OUT/bench_stats.json 2> $OUT/stats.err && \
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py "$@" > /dev/null 2> $OUT/fetch.err && \
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py "$@" > /dev/null 2> $OUT/write.err && \
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
      --output-format csv -d $OUT/sq -- python3 bench.py "$@" > /dev/null 2> $OUT/sq.err && \
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU \
      --output-format csv -d $OUT/sq2 -- python3 bench.py "$@" > /dev/null 2> $OUT/sq2.err
  echo "$OUT exit $?"
  python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
}
COMMON="--steps 4 --warmup 2 --cpu-rays 0 --alt-precision="
run headline_x3mx $COMMON --configs= --precision fp16x3+fp16mx && \
TGTC_BENCH_SINGLE=1 run headline_x3mx_single $COMMON --configs= --precision fp16x3+fp16mx && \
run headline_x3   $COMMON --configs= --precision fp16x3 && \
run styled        $COMMON --configs=styled --precision fp16x3+fp16mx
for d in headline_x3mx headline_x3mx_single headline_x3 styled; do echo "==== $d"; cat $TOP/$d/summary.txt; done
