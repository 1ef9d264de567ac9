#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# scratch: kernel traces of the benchmark for two option sets
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for O in "$@"; do
  i=$((i+1))
  export AA_HIP_OPTIONS="$O"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profab_$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/profab_$i.log 2>&1 || { echo "profile $i failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/profab_$i.log; exit 1; }
  echo "profile $i ($O) done"
done
