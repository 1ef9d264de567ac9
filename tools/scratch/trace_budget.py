# per-iteration kernel time budget from a rocprofv3 kernel trace, iterations delimited by launches of a marker kernel
# usage: trace_budget.py <dir with *_kernel_trace.csv> <marker kernel prefix> [n_iterations_from_the_middle]
import csv, glob, sys, collections
d, marker = sys.argv[1], sys.argv[2]
want = int(sys.argv[3]) if len(sys.argv) > 3 else 60
f = sorted(glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void aa::", "").replace("aa::", "")
names = [short(r["Kernel_Name"]) for r in rows]
marks = [i for i, nm in enumerate(names) if nm.startswith(marker)]
# the longest run of consecutive marker-delimited iterations with the same launch count
runs, cur = [], [marks[0]]
for a, b in zip(marks, marks[1:]):
    if cur and len(cur) > 1 and (b - a) != (cur[-1] - cur[-2]):
        runs.append(cur); cur = [a]
    cur.append(b)
runs.append(cur)
best = max(runs, key=len)
mid = len(best) // 2
sel = best[max(0, mid - want // 2): mid + want // 2 + 1]
a, b, nit = sel[0], sel[-1], len(sel) - 1
tot = collections.OrderedDict(); cnt = collections.Counter()
for i in range(a, b):
    dur = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    tot[names[i]] = tot.get(names[i], 0.0) + dur; cnt[names[i]] += 1
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 / nit
print("marker %s: %d iterations, %.1f us per iteration wall (kernel sum %.1f us), %d launches per iteration"
      % (marker, nit, span, sum(tot.values()) / nit, (b - a) // nit))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-44s %5.2f x %8.1f us = %8.1f us" % (k[:44], cnt[k] / nit, v / cnt[k], v / nit))
