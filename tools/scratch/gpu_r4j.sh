#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for opts in "TWO_RANK_OUTER=1" "TWO_RANK_OUTER=2" "TWO_RANK_OUTER=2 AA_HIP_OPTIONS=proj_mode=1" "TWO_RANK_OUTER=2 AA_HIP_OPTIONS=proj_check=1" "TWO_RANK_OUTER=6 AA_HIP_OPTIONS=qp_mode=1"; do
  echo "######## $opts"
  timeout -k 10 200 python tools/p2p_two_ranks.py TWO_RANK_ONLY_MAIN=1 $opts 2>&1 | grep -v "^$" | grep "ranks vs 1\|costs per\|Error\|MULTI_RANK\|exit\|assert" 
done 2>&1 | tee gpurun_out/r4j_debug.txt
