#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "qp or headline or timed_kernel or hand_over or nonmonotone or medium or converged or restarts or slots" > gpurun_out/r4x_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4x_tests.log
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "qp_quad_lazy=0" "" "qp_quad_lazy=0" | tee gpurun_out/r4x_ab.txt
export BENCH_ARGS="--no-f64"
bash tools/gpu_ab.sh "" "qp_quad_lazy=0" | tee -a gpurun_out/r4x_ab.txt
