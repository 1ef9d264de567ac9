#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# scratch: kernel traces of the benchmark for several option sets + per-iteration budget
mkdir -p gpurun_out
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
i=0
for O in "$@"; do
  i=$((i+1))
  export AA_HIP_OPTIONS="$O"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$i.log 2>&1 || { echo "profile $i failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$i.log; exit 1; }
  echo "=== profile $i ($O)"
  python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$i 30 | tee $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$i.summary
done
