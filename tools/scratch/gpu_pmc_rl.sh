#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# scratch: SQ counters of the row-local kernels (register-staged vs LDS-DMA), one --pmc pass per group
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcrl_$i -- python3 $GRAFT_REPO_ROOT/tools/rl_loop.py > $GRAFT_REPO_ROOT/gpurun_out/pmcrl_$i.log 2>&1
  echo "pmc group $i exit=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcrl_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void aa::", "")[:44]
        if "row_local" in k or "reduce_rows_f32" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(agg)
ctrs = sorted(set(c for k in names for c in agg[k]))
print("%-28s" % "counter", " ".join("%22s" % k[-22:] for k in names))
for c in ctrs:
    print("%-28s" % c, " ".join("%22.0f" % (sum(agg[k][c]) / max(1, len(agg[k][c]))) for k in names))
PY
