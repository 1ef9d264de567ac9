#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# scratch: A/B of option sets on the benchmark in one box; usage: gpu_ab.sh "opts A" "opts B" ...
mkdir -p gpurun_out
i=0
for O in "$@" "$1"; do
  i=$((i+1))
  AA_HIP_OPTIONS="$O" timeout -k 10 300 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { echo "bench failed ($O)"; tail -5 gpurun_out/ab_$i.err; exit 1; }
  python3 -c "
import json; b=json.load(open('gpurun_out/ab_$i.json')); print('%-40s %.1f it/s  %.3f ms' % ('$O' or '(defaults)', b['value'], b['ms_per_step']))"
done
