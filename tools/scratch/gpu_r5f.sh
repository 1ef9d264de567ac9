#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for W in 0 300 0 300 1000 0; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-f64 --clock-warmup-ms $W > gpurun_out/cw_$W.json 2> gpurun_out/cw.err || { tail -3 gpurun_out/cw.err; exit 1; }
  python3 -c "
import json; b=json.load(open('gpurun_out/cw_$W.json')); print('clock warm-up %5s ms: %.1f it/s  %.3f ms' % ('$W', b['value'], b['ms_per_step']))"
done
