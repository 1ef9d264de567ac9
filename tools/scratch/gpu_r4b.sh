#!/bin/bash
# round 4, call B: the re-based parity tests with their yardsticks printed
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_headline.py -q -s -m gpu --timeout 400 -p no:cacheprovider > gpurun_out/r4b_headline.log 2>&1
echo "headline exit=$?"; grep -E "^headline|aa_iterate|passed|failed|Error|assert" gpurun_out/r4b_headline.log | tail -30
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -s -m gpu --timeout 400 -p no:cacheprovider \
    -k "transform_golden or traces_golden or estimator_known or c3_jra55_shape_fixed" > gpurun_out/r4b_yard.log 2>&1
echo "yardsticks exit=$?"; grep -E "^gpnh|^iterate_aa|^aa estimator|^C3|^    weights|passed|failed|Error|^E  " gpurun_out/r4b_yard.log | tail -120
