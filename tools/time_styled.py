#!/usr/bin/env python3
"""Development timing: the stylised ray path, whole 400x400 frame, 128c+64f (TGTC_LIB selects a development build)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for prec in sys.argv[1:] or ["fp16x3"]:
    r = bench.bench_frame(prec, 400, 400, 3, 1, styled=True)
    print("styled %-8s %7.2f ms/frame kernel %7.2f ms  %8.0f rays/s frac %.3f" % (
        prec, r["ms_per_step"], r["kernel_ms"], r["value"], r["roofline"]["frac"]), flush=True)
