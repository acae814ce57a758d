#!/bin/bash
# usage (on the GPU box): tools/bench_variants.sh tag1 tag2 ...   -> one line per dev library
for t in "$@"; do
  TGTC_LIB=$PWD/tgtc-style_amd/csrc/libtgtc_dev_$t.so timeout -k 10 120 python bench.py --steps 4 --warmup 2 --precision ${PREC:-fp16} --alt-precision '' --cpu-rays 0 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$t', 'rays/s %.0f' % d['value'], 'ms/frame %.2f' % d['ms_per_step'], 'fine kernel ms %.2f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'])
except Exception as e: print('$t', 'FAILED', e)"
done
