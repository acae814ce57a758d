#!/bin/bash
# round-2 experiment 16: fp16mx with ONE accumulator chain per row tile (fp6 products accumulate into the fp16 chain)
L=$PWD/tgtc-style_amd/csrc
TGTC_LIB=$L/libtgtc_dev_one.so python -m pytest tests/test_hip_nerf.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  TGTC_LIB=$L/libtgtc_dev_one.so python tools/time_fused.py fp16x3+fp16mx 2>/dev/null | sed 's/^/ONE /'
  python tools/time_fused.py fp16x3+fp16mx 2>/dev/null | sed 's/^/TWO /'
done
