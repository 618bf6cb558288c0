"""Reads a rocprofv3 --kernel-trace CSV of a band run and prints every kernel of one
two-iteration step (from one fused2 launch to the next) with times relative to that launch."""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
f2 = [r for r in rows if "fused2_iteration_kernel" in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s0, s1 = f2[-back][0], f2[-back + 1][0]
lead = 400_000
for a, b, n, q in rows:
    if s0 - lead <= a < s1:
        short = n.split("(")[0].replace("void ", "")[-52:]
        print("%+9.1f us .. %+9.1f us  (%7.1f)  q%-3s %s" % ((a - s0) / 1e3, (b - s0) / 1e3, (b - a) / 1e3, q, short))
