#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider --durations=12 > gpurun_out/r4n_all.log 2>&1
echo "all exit=$?"; tail -25 gpurun_out/r4n_all.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
