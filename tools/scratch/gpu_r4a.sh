#!/bin/bash
# round 4, call A: the re-based parity tests with their yardsticks printed, an A/B of the
# memory-1 instantiation of the wave-per-sample QP kernel at the driver's flags, the whole GPU suite
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_headline.py -q -s -m gpu --timeout 400 -p no:cacheprovider > gpurun_out/r4a_headline.log 2>&1
echo "headline exit=$?"; grep -E "^headline|aa_iterate|passed|failed|Error|assert" gpurun_out/r4a_headline.log | tail -30
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -s -m gpu --timeout 400 -p no:cacheprovider \
    -k "transform_golden or traces_golden or estimator_known or c3_jra55_shape_fixed" > gpurun_out/r4a_yard.log 2>&1
echo "yardsticks exit=$?"; grep -E "^gpnh|^iterate_aa|^aa estimator|^C3|^    weights|passed|failed|Error|assert" gpurun_out/r4a_yard.log | tail -80
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "qp_wave_mem1=0" "qp_wave_mem1=1" || exit 1
timeout -k 10 600 python -m pytest tests -q -m gpu --timeout 400 -p no:cacheprovider -x > gpurun_out/r4a_all.log 2>&1
echo "all exit=$?"; tail -5 gpurun_out/r4a_all.log
