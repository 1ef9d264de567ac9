#!/bin/bash
# copies what tools/gpu_round4_final.sh left under gpurun_out/ into profiles/ (the tracked summaries)
cd "$(dirname "$0")/.."
R=${1:-round4}
cp gpurun_out/final_bench.json profiles/${R}_bench.json
cp gpurun_out/final_bench_s20.json profiles/${R}_bench_steps20.json
cp gpurun_out/final_bench_f64.json profiles/${R}_bench_float64.json
cp gpurun_out/final_bench_n12500.json profiles/${R}_bench_n12500.json
cp gpurun_out/final_prof_f32.summary profiles/${R}_f32_per_iteration.txt
cp gpurun_out/final_prof_f64.summary profiles/${R}_f64_per_iteration.txt
cp "$(ls -t gpurun_out/final_prof_f32/*/*kernel_stats.csv | head -1)" profiles/${R}_f32_kernel_stats.csv
cp "$(ls -t gpurun_out/final_prof_f64/*/*kernel_stats.csv | head -1)" profiles/${R}_f64_kernel_stats.csv
cp "$(ls -t gpurun_out/final_prof_configs/*/*kernel_stats.csv | head -1)" profiles/${R}_configs_kernel_stats.csv
cp gpurun_out/final_configs.jsonl profiles/${R}_configs_c2_c3.jsonl
cp gpurun_out/final_restarts.log profiles/${R}_restarts.txt
[ -f gpurun_out/pmc_traffic.json ] && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
[ -f gpurun_out/pmc_mfma.json ] && cp gpurun_out/pmc_mfma.json profiles/pmc_mfma.json
ls -la profiles/${R}_bench*.json
