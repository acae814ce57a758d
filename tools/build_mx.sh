#!/bin/bash
# Rebuild only the fp16+fp6 translation unit and relink (development loop); extra flags in $1
set -e
cd "$(dirname "$0")/../tgtc-style_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on $1 -c mlp_nerf_mx.hip -o mlp_nerf_mx.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning|VGPRs:|Spill:|ScratchSize" | sort | uniq -c
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ${OUT:-libtgtc_hip.so} common.o raypath.o mlp_nerf.o mlp_nerf_fp16.o mlp_nerf_mx.o render.o mlp_style.o mlp_style_fp16.o style2d.o
echo built
