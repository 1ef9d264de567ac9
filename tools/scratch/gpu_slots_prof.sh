#!/bin/bash
# rocprofv3 kernel statistics of the AA restart slots on the C2 stand-in (tools/aa_slots_check.py 12)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_slots
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_slots -- python3 $GRAFT_REPO_ROOT/tools/aa_slots_only.py > $GRAFT_REPO_ROOT/gpurun_out/prof_slots.log 2>&1 || exit 1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_slots/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_slots_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_slots
