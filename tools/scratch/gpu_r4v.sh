#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "rccl or two_ranks or distributed_mode or shard" > gpurun_out/r4v_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r4v_tests.log
timeout -k 10 300 python3 tools/forced_rccl_bench.py 12500 2>&1 | grep "n=12500" | tee gpurun_out/r4v_forced.txt
AA_HIP_OPTIONS=pack_comm=0 timeout -k 10 300 python3 tools/forced_rccl_bench.py 12500 2>&1 | grep "n=12500" | sed 's/^/pack_comm=0: /' | tee -a gpurun_out/r4v_forced.txt
timeout -k 10 300 python3 tools/forced_rccl_bench.py 2>&1 | grep "n=100000" | tee -a gpurun_out/r4v_forced.txt
timeout -k 10 600 python3 tools/p2p_two_ranks.py > gpurun_out/r4v_two_ranks.txt 2>&1; echo "two ranks rc=$?"; tail -12 gpurun_out/r4v_two_ranks.txt
