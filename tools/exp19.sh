#!/bin/bash
# round-2 experiment 19: the builtin LDS-DMA with the 1 KiB steps in the immediate offset (one 64-bit address per chunk)
L=$PWD/tgtc-style_amd/csrc
python -m pytest tests/test_hip_nerf.py tests/test_fused_gpu.py tests/test_hip_style.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  python tools/time_fused.py fp16x3+fp16mx fp16x3 fp16 2>/dev/null | grep fused | sed 's/^/NEW /'
  TGTC_LIB=$L/libtgtc_dev_old.so python tools/time_fused.py fp16x3+fp16mx fp16x3 fp16 2>/dev/null | grep fused | sed 's/^/OLD /'
done
python tools/time_styled.py fp16x3 2>/dev/null | grep styled | sed 's/^/NEW /'
TGTC_LIB=$L/libtgtc_dev_old.so python tools/time_styled.py fp16x3 2>/dev/null | grep styled | sed 's/^/OLD /'
