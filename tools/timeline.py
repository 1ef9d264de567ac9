# scratch: timeline (start, duration, queue) of the kernels of ONE outer iteration from a rocprofv3 kernel trace
# usage: timeline.py <dir> [iteration index]
import csv, glob, sys
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void aa::", "").replace("aa::", "")[:40]
marks = [i + 1 for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_aa_cost") and i + 1 < len(rows)]
a, b = marks[which], marks[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
