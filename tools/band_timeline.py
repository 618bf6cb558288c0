"""Reads a rocprofv3 --kernel-trace CSV of a band run and prints every kernel of one step with
times relative to the step's first launch of the two-iteration kernel.

  band_timeline.py TRACE_DIR [STEPS_BACK] [LONG_LAUNCHES_PER_STEP]

A step is counted by the long launches of the two-iteration kernel: one per step with one halo row,
two with the deep halo (whose first / last four rows are short launches of the same kernel)."""
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
longest = max(b - a for a, b, _, _ in f2)
f2 = [r for r in f2 if r[1] - r[0] >= longest // 3]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s0, s1 = f2[-back * per_step][0], f2[-(back - 1) * per_step][0]
lead = 100_000
for a, b, n, q in rows:
    if s0 - lead <= a < s1:
        short = n.split("(")[0].replace("void ", "")[-52:]
        print("%+9.1f us .. %+9.1f us  (%7.1f)  q%-3s %s" % ((a - s0) / 1e3, (b - s0) / 1e3, (b - a) / 1e3, q, short))
print("step: %.1f us" % ((s1 - s0) / 1e3))
