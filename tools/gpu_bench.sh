#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-x}
timeout -k 10 500 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?; echo "bench exit=$rc"; if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench_$TAG.err; exit 1; fi
python3 -c "
import json; b=json.load(open('gpurun_out/bench_$TAG.json'))
print({k:b[k] for k in ('value','ms_per_step','mfma_frac_outer_iteration')}); print(b['roofline']); print(b['qp']); print(b.get('parity_converged'))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit=$?"
