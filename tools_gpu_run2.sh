#!/bin/bash
# scratch helper for a gpurun call (not part of the product)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 300 -p no:cacheprovider \
  -k "gram_products or qp_sizes or wide_k or medium or gpnh_golden or aa_estimator" > gpurun_out/t2.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/t2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out; stopping"; exit 1; fi
timeout -k 10 500 python bench.py --steps 10 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err
rc=$?; echo "bench exit=$rc"
if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench1.err; exit 1; fi
cat gpurun_out/bench1.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
echo "rocprof exit=$?"
find $GRAFT_REPO_ROOT/gpurun_out/prof1 -name "*stats*" | head
