#!/bin/bash
# round 4, call E: multi-rank code path on one GPU (forced 1-rank RCCL): device-side overflow handling
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu --timeout 400 -p no:cacheprovider -x \
   -k "rccl or distributed or transform_golden or two_ranks or projection" > gpurun_out/r4e_tests.log 2>&1
echo "tests exit=$?"; tail -8 gpurun_out/r4e_tests.log
for n in 12500 25000 100000; do
  for o in "proj_check=0" "proj_check=1"; do
    echo "== n=$n $o"; AA_HIP_OPTIONS="$o" timeout -k 10 300 python tools/forced_rccl_bench.py $n 2>&1 | tail -2
  done
done | tee gpurun_out/r4e_forced_rccl.txt
