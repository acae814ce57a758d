#!/bin/bash
# One development build of the library with some translation units recompiled under extra flags:
#
#   tools/build_variant.sh <tag> "<extra hipcc flags>" [file.hip ...]      -> tgtc-style_amd/csrc/libtgtc_dev_<tag>.so
#
# Only the listed translation units are recompiled (default: mlp_nerf_mx.hip render_fused.hip); every other object is the
# product build's (run `make -C tgtc-style_amd/csrc` first).  Load the result with TGTC_LIB=<that file>; time two builds
# against each other on ONE box with tools/exp.sh.  Examples (the round-1/2 experiments, profiles/r*_kernel_variants.md):
#   tools/build_variant.sh d2      "-DTGTC_MX_DEPTH=2"                         # fp16mx groups staged two ahead
#   tools/build_variant.sh abl4    "-DTGTC_ABL=4" mlp_nerf.hip mlp_nerf_mx.hip # timing ablation: no LDS fragment reads
#   tools/build_variant.sh p44     "-DTGTC_DEV_VARIANT=MlpCfg<4,4,false,4,kRingSlots,1,true>" mlp_nerf.hip
#   tools/build_variant.sh noslab  "-DTGTC_ABL=16" mlp_style.hip
#   tools/build_variant.sh stag2   "-DTGTC_STAGGER=2"                          # round 3: waves 4-7 two groups behind
set -e
cd "$(dirname "$0")/../tgtc-style_amd/csrc"
TAG="$1"; EXTRA="$2"; shift 2 || true
FILES="${@:-mlp_nerf_mx.hip render_fused.hip}"
OBJS="common.o raypath.o mlp_nerf.o mlp_nerf_fp16.o mlp_nerf_mx.o mlp_nerf_mx2.o mlp_nerf_x3s.o render.o render_fused.o render_styled_fused.o mlp_style.o mlp_style_fp16.o style2d.o mlp_train.o"
pids=""
for f in $FILES; do
  o=/tmp/${f%.hip}_$TAG.o
  OBJS="${OBJS/${f%.hip}.o/$o}"
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on $EXTRA -c $f -o $o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|Spill|ScratchSize" | sort | uniq -c | sed "s/^/$f: /" ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libtgtc_dev_$TAG.so $OBJS
echo built libtgtc_dev_$TAG.so
