#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python tools/sweep_reduce_rows.py > gpurun_out/sweep_rr.log 2>&1; echo "sweep exit=$?"; cat gpurun_out/sweep_rr.log
