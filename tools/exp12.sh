python -m pytest tests/test_hip_style2d.py -x -q -m gpu 2>&1 | tail -3
for ns in 1 2 3 4 6 8 0; do
  TGTC_S2D_NSPLIT=$ns python bench.py --steps 6 --warmup 2 --cpu-rays 0 --alt-precision "" --configs style2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['style2d']; print('NSPLIT=$ns', round(c['value'],3), {k[:8]:round(v,3) for k,v in c['parts_ms'].items()})"
done
