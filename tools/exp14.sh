#!/bin/bash
# round-2 experiment 14: LDS-DMA of the fp16mx streams addressed as SGPR base + 32-bit lane offset + immediate
# (no 64-bit VALU add per DMA instruction) vs per-lane 64-bit pointers
L=$PWD/tgtc-style_amd/csrc
python -m pytest tests/test_hip_nerf.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  python tools/time_fused.py fp16x3+fp16mx 2>/dev/null | sed 's/^/NEW /'
  TGTC_LIB=$L/libtgtc_dev_old.so python tools/time_fused.py fp16x3+fp16mx 2>/dev/null | sed 's/^/OLD /'
done
