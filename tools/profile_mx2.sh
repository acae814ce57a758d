#!/bin/bash
# Kernel times and SQ counters of the per-sample kernels on the split path of the headline frame: the two-tile persistent kernels
# (mlp_nerf_mx2.hip, mlp_nerf_x3s.hip: product) and, with a libtgtc_dev_mx1.so beside it (tools/build_variant.sh mx1
# "-DTGTC_MX2=0 -DTGTC_X3S=0 -I." mlp_nerf_mx.hip mlp_nerf.hip), the one-tile kernels they replaced.  Separate passes for the trace and each counter set (never --pmc with a trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_mx2; rm -rf $OUT; mkdir -p $OUT; cd $R
run() {   # tag, extra rocprofv3 args
  rocprofv3 $2 --output-format csv -d $OUT/$1 -- python3 tools/time_fused.py fp16x3+fp16mx > $OUT/$1.log 2> $OUT/$1.err
}
for lib in product mx1; do
  if [ $lib = mx1 ]; then    # (one development build with BOTH one-tile kernels: -DTGTC_MX2=0 on mlp_nerf_mx.hip, -DTGTC_X3S=0 on mlp_nerf.hip)
    [ -f $R/tgtc-style_amd/csrc/libtgtc_dev_mx1.so ] || continue
    export TGTC_LIB=$R/tgtc-style_amd/csrc/libtgtc_dev_mx1.so
  fi
  run ${lib}_trace "--kernel-trace --stats"
  run ${lib}_a "--pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  run ${lib}_b "--pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"
  run ${lib}_c "--pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for lib in ("product", "mx1"):
    for p in glob.glob(os.path.join(root, lib + "_trace/**/*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(p, newline="")):
            if "nerf" in r["Name"] or "fused" in r["Name"]:
                print("%-8s %-70s calls %5s  avg %10.3f ms" % (lib, r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, lib + "_[abc]/**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            if "nerf_mx" in r["Kernel_Name"] or "nerf_x3s" in r["Kernel_Name"] or "nerf_mlp_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(lib, k)
        wc = sum(cs.get("SQ_WAVE_CYCLES", [1])) / max(len(cs.get("SQ_WAVE_CYCLES", [1])), 1)
        for c, v in sorted(cs.items()):
            m = sum(v) / len(v)
            print("   %-28s mean %.5g  (%.3f of wave cycles)  n=%d" % (c, m, m / wc, len(v)))
PY
