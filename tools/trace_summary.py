# scratch: per-outer-iteration kernel time budget from a rocprofv3 kernel trace of bench.py
# usage: trace_summary.py <dir with *_kernel_trace.csv> [n_iterations_from_the_end]
import csv, glob, sys, collections
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0].replace("void aa::", "").replace("aa::", "")
    return n
names = [short(r["Kernel_Name"]) for r in rows]
# iteration boundaries: the launch after the cost kernel that ends an outer iteration (the dictionary set-up used
# to be a launch of its own and the marker; it is block 0 of the first gradient launch now)
marks = [i + 1 for i, nm in enumerate(names) if nm.startswith("k_aa_cost") and i + 1 < len(names)]
if not marks:
    marks = [i for i, nm in enumerate(names) if nm.startswith("k_dict_setup") or nm.startswith("k_scale_gram")]
if len(sys.argv) > 3:        # [n] [skip]: n iterations after the first `skip` of the trace (the timed window)
    skip = int(sys.argv[3]); marks = marks[skip:skip + last + 1]
else:
    marks = marks[-(last + 12):-11] if len(marks) > last + 12 else marks[:-1]   # skip the 10 timed-gemm iterations at the end
if len(marks) < 3:
    print("too few iterations in trace"); sys.exit(0)
a, b = marks[0], marks[-1]
nit = len(marks) - 1
tot = collections.OrderedDict(); cnt = collections.Counter()
for i in range(a, b):
    dur = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    tot[names[i]] = tot.get(names[i], 0.0) + dur; cnt[names[i]] += 1
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 / nit
print("%d iterations, %.1f us per iteration wall (kernel sum %.1f us), %d launches per iteration"
      % (nit, span, sum(tot.values()) / nit, (b - a) // nit))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-44s %5.2f x %8.1f us = %8.1f us" % (k[:44], cnt[k] / nit, v / cnt[k], v / nit))
