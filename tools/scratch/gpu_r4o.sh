#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_longrun.py tests/test_gpu_headline.py -q -m gpu --timeout 400 -p no:cacheprovider -x -k "qp or medium or converged or timed_kernel or hand_over or nonmonotone" > gpurun_out/r4o_tests.log 2>&1
echo "tests exit=$?"; tail -6 gpurun_out/r4o_tests.log
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "qp_wave_lazy=0" "qp_wave_lazy=1" | tee gpurun_out/r4o_ab.txt
