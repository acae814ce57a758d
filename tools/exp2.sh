#!/bin/bash
# round-2 experiment 2: timing ablations (results are garbage by construction) of the parked 4x4 and the shipped 8x2 fp16 geometries
export TGTC_BENCH_NOCHECK=1
tools/bench_variants.sh p44a0 p44a1 p44a2 p44a3 p44a4 p44a8 p44a15 w82a0 w82a1 w82a2 w82a3 w82a4 w82a8 w82a15
