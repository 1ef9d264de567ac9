#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 300 -p no:cacheprovider -x > gpurun_out/t_r6.log 2>&1
rc=$?; tail -3 gpurun_out/t_r6.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 400 python tools_tune.py 2>&1 | tee gpurun_out/tune2.log
