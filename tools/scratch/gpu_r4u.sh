#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/forced_rccl_bench.py 12500 2>&1 | tee gpurun_out/r4u_forced.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4u -- python3 $GRAFT_REPO_ROOT/bench.py --n 12500 --no-cpu-baseline --no-f64 --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r4u.log 2>&1 || { echo "profile failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_r4u.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/trace_summary.py gpurun_out/prof_r4u 30 | head -30
python3 tools/timeline.py gpurun_out/prof_r4u 30 > gpurun_out/prof_r4u_timeline30.txt
cd /tmp
AA_FORCE_RCCL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4u_rccl -- python3 $GRAFT_REPO_ROOT/tools/forced_rccl_bench.py 12500 > $GRAFT_REPO_ROOT/gpurun_out/prof_r4u_rccl.log 2>&1 || { echo "profile 2 failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_r4u_rccl.log; exit 1; }
