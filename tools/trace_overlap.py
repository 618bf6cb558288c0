"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last few fused-iteration launches,
which other kernels started while each was running (the band's boundary chain on the side
stream and RCCL's transfers) -- evidence of the overlap, or of its absence."""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
fused = [r for r in rows if "fused_iteration_kernel" in r[2] and ", false>" in r[2]]
last = fused[-6:-1]
for k, (s, e, name) in enumerate(last):
    inside = [(a - s, b - s, n) for a, b, n in rows if s <= a < e and "fused_iteration_kernel" not in n]
    nxt = fused[fused.index((s, e, name)) + 1]
    print("fused kernel %d: %.1f us; next one starts %.1f us after this one ends" % (k, (e - s) / 1e3, (nxt[0] - e) / 1e3))
    for a, b, n in inside:
        short = n.split("(")[0][-48:]
        print("    +%7.1f us .. +%7.1f us  %s" % (a / 1e3, b / 1e3, short))
