#!/bin/bash
# round-2 experiment 8: the wide (32x32x16, one wave per SIMD) fp16x3 per-sample kernels vs the 16x16x32 ones, through the
# chain of per-sample kernels (tools/time_fused.py "chain" rows; TGTC_NERF_NARROW=1 forces the 16x16x32 kernels)
python -m pytest tests/test_hip_nerf.py -x -q -m gpu -k "wide" 2>&1 | tail -15
for i in 1 2; do
  TGTC_NERF_WIDE=1 python tools/time_fused.py fp16x3 2>/dev/null | grep chain | sed "s/^/WIDE   /"
  python tools/time_fused.py fp16x3 2>/dev/null | grep chain | sed "s/^/NARROW /"
done
