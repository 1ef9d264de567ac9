#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# MFMA-utilisation counters of the pass kernels and the QP kernels on the headline problem:
# separate --pmc passes (SQ has 8 slots; GRBM is independent), program directly after `--`,
# --kernel-trace only (no sys/hip trace domains together with --pmc).  Summary -> gpurun_out/pmc_mfma.json
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcmfma_$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-f64 > $GRAFT_REPO_ROOT/gpurun_out/pmcmfma_$i.log 2>&1
  echo "pmc group $i ($C) exit=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcmfma_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void aa::", "").replace("aa::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmcmfma_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void aa::", "").replace("aa::", "")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"source": "rocprofv3 --pmc (two passes, tools/gpu_pmc_mfma.sh) on bench.py --steps 12 --warmup 5, MI355X; counter values are per dispatch, "
                 "summed over the chip by rocprofv3; durations from the kernel trace of the first pass (counter collection serialises and slows dispatches)",
       "simds": 1024, "kernels": {}}
for k in sorted(agg):
    if not any(s in k for s in ("k_reduce_rows_f32", "k_row_local_f32", "k_qp_quad", "k_qp_wave", "k_grad")):
        continue
    d = {c: sum(v) / len(v) for c, v in agg[k].items()}
    d["dispatches"] = max(len(v) for v in agg[k].values())
    if dur.get(k):
        d["duration_us_under_pmc"] = sum(dur[k]) / len(dur[k])
    busy, gui = d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("GRBM_GUI_ACTIVE")
    if busy and gui:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of one XCD = gui / 8; MFMA pipes = 1024 SIMDs
        d["mfma_busy_frac_of_simd_cycles"] = busy / ((gui / 8.0) * 1024.0)
    out["kernels"][k] = d
    print("==", k)
    for c, v in sorted(d.items()):
        print("   %-36s %18.4f" % (c, v))
json.dump(out, open("gpurun_out/pmc_mfma.json", "w"), indent=1)
PY
