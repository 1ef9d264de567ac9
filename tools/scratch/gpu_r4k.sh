#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/p2p_two_ranks.py > gpurun_out/r4k_two_ranks.txt 2>&1; echo "exit=$?"; grep -v "^$" gpurun_out/r4k_two_ranks.txt | tail -60
