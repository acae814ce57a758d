#!/bin/bash
# round-2 experiment 17: fp16mx weight groups staged two ahead (second register buffer) instead of one
L=$PWD/tgtc-style_amd/csrc
TGTC_LIB=$L/libtgtc_dev_d2.so python -m pytest tests/test_hip_nerf.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  TGTC_LIB=$L/libtgtc_dev_d2.so python tools/time_fused.py fp16x3+fp16mx 2>/dev/null | sed 's/^/DEPTH2 /'
  python tools/time_fused.py fp16x3+fp16mx 2>/dev/null | sed 's/^/DEPTH1 /'
done
