#!/bin/bash
# round-2 experiment 1: 4-wave AGPR-parked geometry vs the shipped kernels, same box
set -o pipefail
L=$PWD/tgtc-style_amd/csrc
TGTC_LIB=$L/libtgtc_dev_p44.so PREC=fp16 timeout -k 10 300 python tests/probes/check_dev.py && \
TGTC_LIB=$L/libtgtc_dev_p42x3.so PREC=fp16x3 timeout -k 10 300 python tests/probes/check_dev.py && \
tools/bench_variants.sh p44 && PREC=fp16x3 tools/bench_variants.sh p42x3 && \
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --precision fp16 --alt-precision fp16x3 --cpu-rays 0
