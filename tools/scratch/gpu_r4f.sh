#!/bin/bash
# round 4, call F: FurthestSum on the device (tests, restart wall clock), shard iteration direct / multi-rank path
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu --timeout 400 -p no:cacheprovider -x \
   -k "furthest or estimator or restart or transform_golden" > gpurun_out/r4f_tests.log 2>&1
echo "tests exit=$?"; tail -8 gpurun_out/r4f_tests.log
timeout -k 10 400 python tools/slots_check.py 100 > gpurun_out/r4f_slots_check_100.txt 2>&1; tail -12 gpurun_out/r4f_slots_check_100.txt
timeout -k 10 400 python tools/aa_slots_check.py 100 > gpurun_out/r4f_aa_slots_check_100.txt 2>&1; tail -8 gpurun_out/r4f_aa_slots_check_100.txt
for n in 12500 25000; do timeout -k 10 300 python tools/forced_rccl_bench.py $n 2>&1 | grep "n="; done | tee gpurun_out/r4f_forced_rccl.txt
