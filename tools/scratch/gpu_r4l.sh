#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AA_HIP_OPTIONS="row_local_early=1" timeout -k 10 400 python -m pytest tests/test_gpu_longrun.py tests/test_gpu_headline.py -q -m gpu --timeout 300 -p no:cacheprovider -x -k "pass_kernels or timed_kernel_mix" > gpurun_out/r4l_tests.log 2>&1
echo "tests exit=$?"; tail -5 gpurun_out/r4l_tests.log
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "row_local_early=1" "row_local_early=1,row_local_prio=1" | tee gpurun_out/r4l_ab.txt
python3 -c "
import json
for i in (1,2,3,4):
    b=json.load(open('gpurun_out/ab_%d.json'%i)); print(i, b['roofline']['ms_reduce_rows'], b['roofline']['ms_row_local'], b['estimator_loop']['value'])"
