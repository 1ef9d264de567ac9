# scratch: the kernel sequence of one iteration from a rocprofv3 kernel trace (dir, anchor kernel name)
import csv, glob, sys
d, anchor = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void aa::", "").replace("aa::", "") for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith(anchor)]
a, b = marks[-3], marks[-2]
t0 = int(rows[a]["Start_Timestamp"])
for i in range(a, b):
    print("%8.1f us  +%6.1f  %s" % ((int(rows[i]["Start_Timestamp"]) - t0) / 1e3,
                                     (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, names[i][:60]))
print("iteration: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
