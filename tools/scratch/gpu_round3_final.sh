#!/bin/bash
# round-3 artefacts in one call: bench lines + rocprofv3 statistics (gpu_final.sh), HBM traffic
# counters (gpu_pmc.sh), MFMA-busy counters (gpu_pmc_mfma.sh)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh > gpurun_out/final_pmc.log 2>&1 || { tail -5 gpurun_out/final_pmc.log; exit 1; }
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json      # bench.py reads it from there
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc_mfma.sh > gpurun_out/final_pmc_mfma.log 2>&1 || { tail -5 gpurun_out/final_pmc_mfma.log; exit 1; }
cd $GRAFT_REPO_ROOT
[ -f gpurun_out/pmc_mfma.json ] && cp gpurun_out/pmc_mfma.json profiles/round3_pmc_mfma.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmcmfma_1 gpurun_out/pmcmfma_2
bash tools/gpu_final.sh
