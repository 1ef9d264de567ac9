#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# round-2 helper: the whole GPU suite, then an A/B of option sets on the benchmark.
# usage: gpu_round2.sh TAG "optsA" "optsB" ...
mkdir -p gpurun_out
TAG=${1:-x}; shift
timeout -k 10 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/t_$TAG.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/t_$TAG.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" gpurun_out/t_$TAG.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out; stopping"; exit 1; fi
i=0
for O in "$@"; do
  i=$((i+1))
  AA_HIP_OPTIONS="$O" timeout -k 10 300 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_${TAG}_$i.json 2> gpurun_out/ab_${TAG}_$i.err || { echo "bench failed ($O)"; tail -5 gpurun_out/ab_${TAG}_$i.err; exit 1; }
  python3 -c "
import json; b=json.load(open('gpurun_out/ab_${TAG}_$i.json')); print('%-40s %.1f it/s  %.3f ms  qp %s' % ('$O' or '(defaults)', b['value'], b['ms_per_step'], b['qp']))"
done
