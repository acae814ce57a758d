#!/bin/bash
# round-2 experiment 7: what the scratch-slab traffic of the stylised kernel costs (slab stores / loads removed, -DTGTC_ABL=16)
L=$PWD/tgtc-style_amd/csrc
for i in 1 2; do
  python tools/time_styled.py fp16x3 fp16 2>/dev/null | grep styled
  TGTC_LIB=$L/libtgtc_dev_noslab.so python tools/time_styled.py fp16x3 2>/dev/null | grep styled | sed 's/^/NOSLAB /'
done
