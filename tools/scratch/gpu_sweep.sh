#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/sweep_row_local.py > gpurun_out/sweep_rl.log 2>&1; echo "sweep exit=$?"; cat gpurun_out/sweep_rl.log
