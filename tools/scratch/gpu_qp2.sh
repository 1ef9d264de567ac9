#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# scratch: the weights QP of the benchmark problem: profile counters, then a kernel trace
mkdir -p gpurun_out
TAG=${1:-x}
QP_PROFILE=1 timeout -k 10 300 python tools/qp_profile.py > gpurun_out/qpp_$TAG.log 2>&1
cat gpurun_out/qpp_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/qpprof_$TAG -- python3 $GRAFT_REPO_ROOT/tools/qp_profile.py > $GRAFT_REPO_ROOT/gpurun_out/qpprof_$TAG.log 2>&1
echo "rocprof exit=$?"
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/qpprof_$TAG/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-50s calls %5s avg %9.1f us tot %8.2f ms"%(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
