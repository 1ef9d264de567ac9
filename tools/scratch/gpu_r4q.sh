#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "qp_fused_order=0" "" "qp_fused_order=0" | tee gpurun_out/r4q_ab.txt
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4q_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4q_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4q -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f64 --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r4q.log 2>&1 || { echo "profile failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_r4q.log; exit 1; }
cd $GRAFT_REPO_ROOT
echo "-- the driver's window (outer iterations 5..25)"; python3 tools/trace_summary.py gpurun_out/prof_r4q 20 5 | head -8
python3 tools/timeline.py gpurun_out/prof_r4q 15 > gpurun_out/prof_r4q_timeline15.txt
