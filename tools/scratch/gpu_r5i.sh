#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "headline or timed_kernel or schedule_knobs or side_by_side or golden or estimator_known or c4_ or restart or rccl" > gpurun_out/r5i_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5i_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f64 --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r5i.log 2>&1 || { echo "profile failed"; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/trace_summary.py gpurun_out/prof_r5i 20 5 | grep "iterations,\|k_gram_wide_pq\|k_linesearch_fin"
