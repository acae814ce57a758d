#!/bin/bash
# round-2 experiment 9: timing ablations of the wide fp16x3 per-sample kernels (chain rows of tools/time_fused.py)
L=$PWD/tgtc-style_amd/csrc
for a in 0 1 2 4 8 15 0; do
  TGTC_LIB=$L/libtgtc_dev_w$a.so python tools/time_fused.py fp16x3 2>/dev/null | grep chain | sed "s/^/ABL$a /"
done
