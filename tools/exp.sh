#!/bin/bash
# A/B timing of library builds on ONE box, interleaved (cdna_hip_programming.md rule 24: never rank builds across boxes):
#
#   tools/exp.sh [-n repeats] [-t "pytest files"] [-c "timer command"] <tag|product> <tag|product> ...
#
# Every round runs the timer once per build (TGTC_LIB=csrc/libtgtc_dev_<tag>.so; `product` = the shipped library); -t first
# runs the named GPU tests against every build.  Default timer: tools/time_fused.py fp16x3+fp16mx (fused ray kernel and the
# per-sample chain, whole 400x400 frame).  Other timers: "python tools/time_styled.py fp16x3", "python tools/time_train.py",
# "python bench.py --steps 6 --warmup 2 --cpu-rays 0 --alt-precision '' --configs style2d".
# This script replaces the one-off tools/exp1.sh ... exp21.sh of rounds 1-2 (git history has them; their results are
# profiles/r1_kernel_variants.md and profiles/r2_kernel_variants.md).
N=3; TESTS=""; CMD="python tools/time_fused.py fp16x3+fp16mx"
while getopts "n:t:c:" o; do case $o in n) N=$OPTARG;; t) TESTS=$OPTARG;; c) CMD=$OPTARG;; esac; done
shift $((OPTIND - 1))
L=$PWD/tgtc-style_amd/csrc
lib() { if [ "$1" = product ]; then echo $L/libtgtc_hip.so; else echo $L/libtgtc_dev_$1.so; fi; }
if [ -n "$TESTS" ]; then
  for t in "$@"; do
    TGTC_LIB=$(lib $t) timeout -k 10 900 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -2 | sed "s/^/[$t] /" || exit 1
  done
fi
for i in $(seq $N); do
  for t in "$@"; do
    TGTC_LIB=$(lib $t) timeout -k 10 300 $CMD 2>/dev/null | sed "s/^/[$t] /"
  done
done
