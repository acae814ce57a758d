#!/bin/bash
# End-of-round check on the GPU box, in three stages of at most one gpurun call each (a call is limited to 20 minutes and the GPU
# suite alone takes 15):
#   tools/round_check.sh tests      whole GPU suite                                   -> gpurun_out/round_check/pytest_gpu.txt
#   tools/round_check.sh bench      smoke, default bench line, two-rank rehearsals (gloo, both ranks on the one GPU)
#   tools/round_check.sh profile    the profile passes of tools/profile_r4.sh        -> gpurun_out/prof_r4/, then on the build machine:
#                                   python tools/refresh_profiles.py gpurun_out/prof_r4 r4
set -o pipefail
O=gpurun_out/round_check; mkdir -p $O
case "${1:-tests}" in
tests)
  python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" | tee -a $O/status.txt; tail -3 $O/pytest_gpu.txt;;
bench)
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" | tee -a $O/status.txt; tail -2 $O/smoke.txt
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" | tee -a $O/status.txt; tail -c 600 $O/bench_default.json
  for sh in frames rays; do
    TGTC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 2 --steps 3 --warmup 1 --sharding $sh --cpu-rays 0 --alt-precision "" --configs "" > $O/bench_2rank_$sh.json 2> $O/bench_2rank_$sh.err
    echo "2-rank $sh rc $?" | tee -a $O/status.txt; tail -c 400 $O/bench_2rank_$sh.json; echo
  done;;
profile)
  bash tools/profile_r4.sh > $O/profile.txt 2>&1; echo "profile rc $?" | tee -a $O/status.txt; tail -60 $O/profile.txt;;
*) echo "usage: tools/round_check.sh tests|bench|profile"; exit 2;;
esac
