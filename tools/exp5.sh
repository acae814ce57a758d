#!/bin/bash
# round-2 experiment 5: fp16x3 per-sample kernels with the LDS-DMA issued through inline asm (counted LDS waits) vs the builtin
export TGTC_BENCH_CHAIN=1 PREC=fp16x3
tools/bench_variants.sh x3blt x3asm x3blt x3asm
