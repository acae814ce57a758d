#!/bin/bash
# round-2 experiment 18: fp16x3 / fp16 passes of the fused ray kernel with asm LDS-DMA + SGPR-base addressing vs the builtin
L=$PWD/tgtc-style_amd/csrc
TGTC_LIB=$L/libtgtc_dev_x3asm.so python -m pytest tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  TGTC_LIB=$L/libtgtc_dev_x3asm.so python tools/time_fused.py fp16x3+fp16mx fp16x3 fp16 2>/dev/null | grep fused | sed 's/^/ASM     /'
  python tools/time_fused.py fp16x3+fp16mx fp16x3 fp16 2>/dev/null | grep fused | sed 's/^/BUILTIN /'
done
