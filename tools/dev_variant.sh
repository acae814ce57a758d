#!/bin/bash
# Build a development copy of the library with only the fp16 ray-mode NeRF kernels of one experimental
# configuration (fast compile), e.g.  tools/dev_variant.sh 'MlpCfg<4,4,false,4>' g4
# Produces tgtc-style_amd/csrc/libtgtc_hip.so.<tag> ; bench with TGTC_LIB=<that file>.
set -e
cd "$(dirname "$0")/../tgtc-style_amd/csrc"
VAR="$1"; TAG="$2"; EXTRA="$3"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on "-DTGTC_DEV_VARIANT=$VAR" $EXTRA -c mlp_nerf.hip -o /tmp/mlp_nerf_$TAG.o -Rpass-analysis=kernel-resource-usage
hipcc -shared -fPIC --offload-arch=gfx950 -o libtgtc_dev_$TAG.so common.o raypath.o /tmp/mlp_nerf_$TAG.o mlp_nerf_mx.o render.o mlp_style.o mlp_style_fp16.o style2d.o render_fused.o
echo built libtgtc_dev_$TAG.so
