#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "pq_mfma=0" "" "pq_mfma=0" | tee gpurun_out/r5g_ab.txt
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r5g_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5g_tests.log
