#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# helper for gpurun calls (run from the repo root): tests, bench, rocprofv3 kernel stats
mkdir -p gpurun_out
TAG=${1:-x}
timeout -k 10 700 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider > gpurun_out/t_$TAG.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/t_$TAG.log; tail -4 gpurun_out/t_$TAG.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out; stopping"; exit 1; fi
timeout -k 10 500 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?; echo "bench exit=$rc"
if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench_$TAG.err; exit 1; fi
cat gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit=$?"
