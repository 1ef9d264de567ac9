#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
# scratch: kernel durations of the weights update, lane+wave (2) against quad+wave (4), early / late
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for st in early late; do for m in 2 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/qp4_${st}_$m -- python3 $GRAFT_REPO_ROOT/tools/qp_quad_tune.py 100000 $st $m > $GRAFT_REPO_ROOT/gpurun_out/qp4_${st}_$m.log 2>&1 || exit 1
  echo "== $st mode $m"; tail -1 $GRAFT_REPO_ROOT/gpurun_out/qp4_${st}_$m.log
  python3 - $GRAFT_REPO_ROOT/gpurun_out/qp4_${st}_$m <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the last 6 weights updates: durations of the QP kernels in launch order
sel = [r for r in rows if any(t in r["Kernel_Name"] for t in ("k_qp<", "k_qp_quad", "k_qp_wave", "k_qp_order", "k_qp_setup"))]
out = {}
for r in sel[-36:]:
    out.setdefault(r["Kernel_Name"].split("(")[0][:40], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in out.items(): print("   %-42s %s us" % (k, " ".join("%.0f" % x for x in v[-6:])))
PY
done; done
