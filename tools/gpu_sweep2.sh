#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/sweep_reduce_rows.py > gpurun_out/sweep_rr.log 2>&1; echo "sweep exit=$?"; cat gpurun_out/sweep_rr.log
