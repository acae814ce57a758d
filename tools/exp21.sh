#!/bin/bash
# round-2 experiment 21: 2-D pass with the GEMM B operands (network weights) pre-split into fp16 hi / lo at handle creation
L=$PWD/tgtc-style_amd/csrc
python -m pytest tests/test_hip_style2d.py tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py --steps 6 --warmup 2 --cpu-rays 0 --alt-precision "" --configs style2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['style2d']; print('PRE', round(c['value'],3), {k[:8]:round(v,3) for k,v in c['parts_ms'].items()})"
  TGTC_LIB=$L/libtgtc_dev_old.so python bench.py --steps 6 --warmup 2 --cpu-rays 0 --alt-precision "" --configs style2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['style2d']; print('OLD', round(c['value'],3), {k[:8]:round(v,3) for k,v in c['parts_ms'].items()})"
done
