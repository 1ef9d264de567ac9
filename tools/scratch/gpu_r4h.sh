#!/bin/bash
# round 4, call H: A/B sweep of small knobs at the driver's flags
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "row_local_prio=1" "row_local_nt=1" "qp_quad_occ=2" "qp_quad_occ=4" "qp_overlap_tail=1" "qp_overlap_tail=1,qp_tail_cap=64" "use_graph=1" "qp_wave_blocks=2048" | tee gpurun_out/r4h_ab.txt
