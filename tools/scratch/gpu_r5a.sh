#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "qp_wave_queue=1,qp_wave_blocks=256" "qp_wave_queue=1,qp_wave_blocks=512" "qp_wave_queue=1" "qp_wave_blocks=256" "" | tee gpurun_out/r5a_ab.txt
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "schedule_knobs or headline" 2>&1 | tail -3
