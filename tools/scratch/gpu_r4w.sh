#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/forced_rccl_bench.py 12500 2>&1 | grep "n=12500" | tee gpurun_out/r4w_forced.txt
timeout -k 10 300 python bench.py --n 12500 --no-cpu-baseline --no-f64 > gpurun_out/r4w_n12500.json 2> gpurun_out/r4w_n12500.err && python3 -c "import json; b=json.load(open('gpurun_out/r4w_n12500.json')); print('n=12500', b['value'], b['ms_per_step'])"
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4w_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4w_tests.log
