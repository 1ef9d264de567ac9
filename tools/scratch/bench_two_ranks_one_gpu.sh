#!/bin/bash
# rehearsal of bench.py --gpus 2 with both ranks on ONE GPU (peer-to-peer transport; RCCL cannot put two ranks on a device)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29617 AA_LAUNCH_ID=$$ AA_COMM=p2p
RANK=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-f64 > gpurun_out/bench2_rank1.log 2>&1 &
P1=$!
RANK=0 timeout -k 10 400 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-f64 > gpurun_out/bench2_rank0.json 2> gpurun_out/bench2_rank0.err
R0=$?
wait $P1; R1=$?
echo "rank0 rc=$R0 rank1 rc=$R1"
python3 -c "
import json; b=json.load(open('gpurun_out/bench2_rank0.json')); print({k:b[k] for k in ('metric','value','n_gpus','ms_per_step','scaling')}); print(b['config']); print('cost_last', b.get('cost_last'))"
tail -3 gpurun_out/bench2_rank1.log
unset WORLD_SIZE RANK LOCAL_RANK AA_COMM
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f64 > gpurun_out/bench1.json 2>/dev/null
python3 -c "
import json; b=json.load(open('gpurun_out/bench1.json')); print('one rank:', b['value'], b['ms_per_step'], 'cost_last', b.get('cost_last'))"
