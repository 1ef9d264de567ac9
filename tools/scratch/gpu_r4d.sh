#!/bin/bash
# round 4, call D: whole GPU suite, A/B of the prefetched QP sample order, kernel trace
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r4d_all.log 2>&1
echo "all exit=$?"; tail -15 gpurun_out/r4d_all.log
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "qp_prefetch_order=0" "qp_prefetch_order=1" "qp_prefetch_order=1,grad_side=0" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4d -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f64 --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r4d.log 2>&1 || { echo "profile failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_r4d.log; exit 1; }
cd $GRAFT_REPO_ROOT
echo "-- the driver's window (outer iterations 5..25)"; python3 tools/trace_summary.py gpurun_out/prof_r4d 20 5 | tee gpurun_out/prof_r4d_window.txt | head -12
echo "-- steady state"; python3 tools/trace_summary.py gpurun_out/prof_r4d 30 | tee gpurun_out/prof_r4d_steady.txt | head -12
python3 tools/timeline.py gpurun_out/prof_r4d 15 > gpurun_out/prof_r4d_timeline15.txt; python3 tools/timeline.py gpurun_out/prof_r4d 45 > gpurun_out/prof_r4d_timeline45.txt
