#!/bin/bash
# round-2 experiment 6: static s_setprio 1 for waves 4-7 of the fused ray kernel (CDNA guide T5, static form)
L=$PWD/tgtc-style_amd/csrc
for i in 1 2; do
  python tools/time_fused.py fp16x3+fp16mx fp16x3 2>/dev/null | grep fused
  TGTC_LIB=$L/libtgtc_dev_prio.so python tools/time_fused.py fp16x3+fp16mx fp16x3 2>/dev/null | grep fused | sed 's/^/PRIO /'
done
