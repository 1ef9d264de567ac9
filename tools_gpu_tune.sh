#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python tools_tune.py 2>&1 | tee gpurun_out/tune.log
