#!/bin/bash
# round 4, call I: peer-to-peer all-reduce -- one rank (bit-identical to RCCL), two ranks on one GPU; forced-path timing
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 200 -p no:cacheprovider -x -k "rccl_path_single_rank or furthest_sum_on_the_device" > gpurun_out/r4i_t1.log 2>&1
echo "single-rank exit=$?"; tail -6 gpurun_out/r4i_t1.log
timeout -k 10 500 python -m pytest tests/test_gpu_configs.py -q -m gpu --timeout 450 -p no:cacheprovider -x -k "two_ranks_on_one_gpu" > gpurun_out/r4i_t2.log 2>&1
echo "two-rank exit=$?"; tail -30 gpurun_out/r4i_t2.log
