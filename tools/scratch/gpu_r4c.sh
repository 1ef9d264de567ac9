#!/bin/bash
# round 4, call C: whole GPU suite, A/B of the side-stream SPG tail and the quad pass cap at the
# driver's flags, kernel trace of the driver's window
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r4c_all.log 2>&1
echo "all exit=$?"; tail -15 gpurun_out/r4c_all.log
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "grad_side=0" "grad_side=1" "qp_quad_cap=24" "qp_quad_cap=48" "qp_quad_cap=16" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f64 --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r4c.log 2>&1 || { echo "profile failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_r4c.log; exit 1; }
cd $GRAFT_REPO_ROOT
echo "-- the driver's window (outer iterations 5..25)"; python3 tools/trace_summary.py gpurun_out/prof_r4c 20 5 | tee gpurun_out/prof_r4c_window.txt
echo "-- steady state"; python3 tools/trace_summary.py gpurun_out/prof_r4c 30 | tee gpurun_out/prof_r4c_steady.txt
