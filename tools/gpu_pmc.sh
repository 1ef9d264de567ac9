#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# scratch: HBM traffic counters for the two GEMM kernels (separate --pmc passes, as the
# MI355X guide prescribes; no sys/hip trace domains together with --pmc)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f64 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
  echo "pmc $C exit=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % C, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    print("==", C, "(KB per dispatch; FETCH_SIZE under-reports wide streaming reads by 2x on gfx950)")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("%-60s n=%4d mean=%12.1f KB" % (k[-60:], len(v), sum(v) / len(v)))
    globals().setdefault("res", {})[C] = {k: sum(v) / len(v) for k, v in agg.items()}
import json
def pick(C, key):
    c = [v for k, v in res[C].items() if key in k]
    return max(c) if c else None
import hashlib, subprocess
sha = hashlib.sha1(open("matrix-factorization-case-studies_amd/csrc/kernels_gemm.hip", "rb").read()).hexdigest()
out = {"n": 100000, "p": 4096, "k": 32, "n_gpus": 1, "dtype": "float32",
       "kernels_gemm_sha1": sha,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_pmc.sh); bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 per dispatch (FETCH_SIZE doubled: gfx950 reports half of 16-B/lane streaming reads, MI355X_MICROARCH.md section HBM; the 8-B/lane operand reads of k_reduce_rows_f32 are uncalibrated)",
       "algorithmic_bytes": 100000 * 4096 * 4}
for name, key in (("reduce_rows", "k_reduce_rows_f32"), ("row_local", "k_row_local_f32")):
    f, w = pick("FETCH_SIZE", key), pick("WRITE_SIZE", key)
    out[name + "_FETCH_SIZE_KB"], out[name + "_WRITE_SIZE_KB"] = f, w
    out[name + "_bytes"] = (2 * f + w) * 1024 if f is not None and w is not None else None
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
