#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# round artefacts: bench lines and rocprofv3 kernel statistics for every configuration measured
# (outputs under gpurun_out/final_*; the ones quoted in DESIGN.md are copied to profiles/)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { echo "== $1" | tee -a gpurun_out/final_progress.log; }
step "bench (default flags)"
timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -5 gpurun_out/final_bench.err; exit 1; }
python3 -c "import json; b=json.load(open('gpurun_out/final_bench.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b.get('estimator_loop'), b.get('float64',{}).get('value'), b['cpu_baseline']['value'])"
step "bench (driver flags: --steps 20 --warmup 5)"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 > gpurun_out/final_bench_s20.json 2> gpurun_out/final_bench_s20.err || exit 1
python3 -c "import json; b=json.load(open('gpurun_out/final_bench_s20.json')); print(b['value'], b['ms_per_step'])"
step "bench float64"
timeout -k 10 300 python bench.py --dtype float64 --no-cpu-baseline > gpurun_out/final_bench_f64.json 2> gpurun_out/final_bench_f64.err || exit 1
python3 -c "import json; b=json.load(open('gpurun_out/final_bench_f64.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'])"
step "bench 12 500-row shard"
timeout -k 10 300 python bench.py --n 12500 --no-cpu-baseline --no-f64 > gpurun_out/final_bench_n12500.json 2> gpurun_out/final_bench_n12500.err || exit 1
python3 -c "import json; b=json.load(open('gpurun_out/final_bench_n12500.json')); print(b['value'], b['ms_per_step'])"
step "12 500-row shard with the row QP kernel (default: the four-lane kernel)"
AA_HIP_OPTIONS=qp_mode=3 timeout -k 10 300 python bench.py --n 12500 --no-cpu-baseline --no-f64 > gpurun_out/final_bench_n12500_row.json 2> gpurun_out/final_bench_n12500_row.err || exit 1
python3 -c "import json; b=json.load(open('gpurun_out/final_bench_n12500_row.json')); print(b['value'], b['ms_per_step'])"
step "C2 / C3 stand-ins"
timeout -k 10 300 python tools/bench_configs.py 200 > gpurun_out/final_configs.jsonl 2> gpurun_out/final_configs.err || { tail -5 gpurun_out/final_configs.err; exit 1; }
cat gpurun_out/final_configs.jsonl
step "restarts"
timeout -k 10 400 python tools/restarts_bench.py 20 > gpurun_out/final_restarts.log 2>&1 || { tail -5 gpurun_out/final_restarts.log; exit 1; }
cat gpurun_out/final_restarts.log
cd /tmp && export TMPDIR=/tmp
step "rocprofv3: bench float32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-f64 > $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32 50 | tee $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32.summary
echo "-- the driver's window (outer iterations 5..25 of the run)" | tee -a $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32.summary
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32 20 5 | tee -a $GRAFT_REPO_ROOT/gpurun_out/final_prof_f32.summary
step "rocprofv3: bench float64"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof_f64 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype float64 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/final_prof_f64.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/final_prof_f64 50 | tee $GRAFT_REPO_ROOT/gpurun_out/final_prof_f64.summary
step "rocprofv3: C2 / C3 stand-ins"
export BENCH_CONFIGS_KERNEL=0   # the profile is of the two small configurations only
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof_configs -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py 100 > $GRAFT_REPO_ROOT/gpurun_out/final_prof_configs.log 2>&1 || exit 1
echo done
