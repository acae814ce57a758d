#!/bin/bash
# Round-2 profiles of the default bench command (run on the GPU box; summaries are copied to profiles/ by hand).
#   1. per-kernel times:  rocprofv3 --kernel-trace --stats
#   2. HBM traffic:       rocprofv3 --pmc FETCH_SIZE  /  --pmc WRITE_SIZE   (separate passes, no trace domains)
#   3. matrix pipe / LDS: rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r2
mkdir -p $OUT
ARGS="--steps 4 --warmup 2 --cpu-rays 0 --alt-precision fp16x3 --configs styled,style2d,trex_rays"
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > /dev/null 2> $OUT/fetch.err && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > /dev/null 2> $OUT/write.err && \
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq -- python3 bench.py $ARGS > /dev/null 2> $OUT/sq.err
echo "exit $?"
find $OUT -name "*.csv" | head -20
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
