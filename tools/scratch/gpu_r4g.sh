#!/bin/bash
# round 4, call G: non-blocking streams -- whole GPU suite, worker threads with and without hipGraph replay, bench A/B
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/r4g_all.log 2>&1
echo "all exit=$?"; tail -12 gpurun_out/r4g_all.log
( timeout -k 10 300 python tools/restarts_threads.py 20; AA_HIP_OPTIONS="use_graph=1" timeout -k 10 300 python tools/restarts_threads.py 20 ) > gpurun_out/r4g_restarts_threads.txt 2>&1
tail -14 gpurun_out/r4g_restarts_threads.txt
timeout -k 10 300 python tools/interleave_probe.py > gpurun_out/r4g_interleave.txt 2>&1; tail -6 gpurun_out/r4g_interleave.txt
export BENCH_ARGS="--steps 20 --warmup 5 --no-f64"
bash tools/gpu_ab.sh "" "qp_prefetch_order=0" || exit 1
