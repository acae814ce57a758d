#!/bin/bash
# development: ablation builds of the wide per-sample kernels: tools/build_wide_abl.sh abl... -> csrc/libtgtc_dev_w<abl>.so
cd "$(dirname "$0")/../tgtc-style_amd/csrc"
for a in "$@"; do
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DTGTC_ABL=$a $EXTRA -c mlp_nerf_wide.hip -o /tmp/mlp_nerf_wide_a$a.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o libtgtc_dev_w$a.so common.o raypath.o mlp_nerf.o /tmp/mlp_nerf_wide_a$a.o mlp_nerf_mx.o render.o render_fused.o mlp_style.o style2d.o mlp_nerf_fp16.o mlp_style_fp16.o && echo built w$a ) &
done
wait
