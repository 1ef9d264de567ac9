#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "qp_overlap_tail=1" "qp_overlap_tail=1,qp_tail_cap=64" "qp_overlap_tail=1,qp_tail_cus=8" "qp_overlap_tail=1,qp_tail_cus=8,qp_tail_cap=64" "qp_overlap_tail=1,qp_tail_cus=16,qp_tail_cap=64" "qp_overlap_tail=1,qp_tail_cus=8,qp_tail_mask_mode=1,qp_tail_cap=64" | tee gpurun_out/r4t_ab.txt
cd /tmp && export TMPDIR=/tmp
AA_HIP_OPTIONS="qp_overlap_tail=1,qp_tail_cus=8,qp_tail_cap=64" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f64 --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r4t.log 2>&1 || { echo "profile failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_r4t.log; exit 1; }
cd $GRAFT_REPO_ROOT
echo "-- the driver's window (outer iterations 5..25)"; python3 tools/trace_summary.py gpurun_out/prof_r4t 20 5 | head -10
echo "-- steady"; python3 tools/trace_summary.py gpurun_out/prof_r4t 30 | head -10
python3 tools/timeline.py gpurun_out/prof_r4t 15 > gpurun_out/prof_r4t_timeline15.txt
python3 tools/timeline.py gpurun_out/prof_r4t 45 > gpurun_out/prof_r4t_timeline45.txt
