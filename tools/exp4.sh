#!/bin/bash
# round-2 experiment 4: fp16mx epilogue without fp32 arithmetic (fma_mix lo halves, integer block max), 8x1 and parked 4x2
set -o pipefail
L=$PWD/tgtc-style_amd/csrc
for t in mx8d mxpd; do
  echo "== $t"
  TGTC_LIB=$L/libtgtc_dev_$t.so PREC=fp16mx timeout -k 10 300 python tests/probes/check_dev.py || exit 1
  TGTC_LIB=$L/libtgtc_dev_$t.so timeout -k 10 300 python bench.py --steps 4 --warmup 2 --precision fp16mx --alt-precision '' --cpu-rays 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', 'rays/s %.0f' % d['value'], 'ms/frame %.2f' % d['ms_per_step'], 'fine kernel ms %.2f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'])" || exit 1
done
TGTC_LIB=$L/libtgtc_dev_mx8d.so timeout -k 10 600 python -m pytest tests/test_hip_nerf.py -x -q -k "mx or mixed" 2>&1 | tail -5
