#!/bin/bash
# build ablation variants in parallel: tools/build_abl.sh <cfg> <tagprefix> abl...
CFG="$1"; PFX="$2"; shift 2
for a in "$@"; do
  ( tools/dev_variant.sh "$CFG" ${PFX}a$a "-DTGTC_ABL=$a" > /tmp/build_${PFX}a$a.log 2>&1; grep -E "error|built" /tmp/build_${PFX}a$a.log ) &
done
wait
