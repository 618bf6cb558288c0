"""Mean of each counter per kernel from rocprofv3 --pmc ... --output-format csv output
(*_counter_collection.csv anywhere under the given directory)."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"][:60], row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
for (kernel, counter), (total, n) in sorted(acc.items()):
    print("%-60s %-28s n=%-4d mean=%.6g" % (kernel, counter, n, total / n))
