#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 300 -p no:cacheprovider -x -k "qp or medium" > gpurun_out/t_r12.log 2>&1
rc=$?; tail -3 gpurun_out/t_r12.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python tools_tune.py 2>&1 | tee gpurun_out/tune4.log
