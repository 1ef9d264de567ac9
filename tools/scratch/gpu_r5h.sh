#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r5h_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5h_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5h -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f64 --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r5h.log 2>&1 || { echo "profile failed"; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/trace_summary.py gpurun_out/prof_r5h 20 5 | grep "iterations,\|k_gram_wide_pq\|k_linesearch_fin"
