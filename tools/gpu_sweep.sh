#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/sweep_row_local.py > gpurun_out/sweep_rl.log 2>&1; echo "sweep exit=$?"; cat gpurun_out/sweep_rl.log
