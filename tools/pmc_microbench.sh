cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r4/mb; mkdir -p $OUT
for b in mx_layer mx_layer_noreads_noepi mx_layer_bare; do
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/${b}_a -- $R/tools/microbench/$b 100 > $OUT/$b.a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/${b}_b -- $R/tools/microbench/$b 100 > $OUT/$b.b.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for b in ("mx_layer", "mx_layer_noreads_noepi", "mx_layer_bare"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, b + "_[ab]/**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(b, k)
        wc = max(cs.get("SQ_WAVE_CYCLES", [1]))
        for c, v in sorted(cs.items()):
            print("   %-28s max %.4g  (%.3f of wave cycles)" % (c, max(v), max(v) / wc))
PY
