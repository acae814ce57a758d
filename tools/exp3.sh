#!/bin/bash
# round-2 experiment 3: parked 4x2 fp16mx kernel vs the shipped 8x1 one
set -o pipefail
L=$PWD/tgtc-style_amd/csrc
TGTC_LIB=$L/libtgtc_dev_mxp.so PREC=fp16mx timeout -k 10 300 python tests/probes/check_dev.py && \
TGTC_LIB=$L/libtgtc_dev_mxp.so timeout -k 10 300 python bench.py --steps 4 --warmup 2 --precision fp16mx --alt-precision 'fp16x3+fp16mx' --cpu-rays 0 && \
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --precision fp16mx --alt-precision '' --cpu-rays 0
