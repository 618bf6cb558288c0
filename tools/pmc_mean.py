"""Mean, minimum and maximum of each counter per kernel from rocprofv3 --pmc ... --output-format csv
output (*_counter_collection.csv anywhere under the given directory).  A counter that rocprofv3
reports once per XCD / shader engine is summed per dispatch first."""
import collections
import csv
import glob
import sys

per_dispatch = collections.defaultdict(float)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"][:60], row["Counter_Name"], row.get("Dispatch_Id", ""))
            per_dispatch[key] += float(row["Counter_Value"])
acc = collections.defaultdict(list)
for (kernel, counter, _), v in per_dispatch.items():
    acc[(kernel, counter)].append(v)
for (kernel, counter), vals in sorted(acc.items()):
    print("%-60s %-28s n=%-4d mean=%.6g min=%.6g max=%.6g" % (kernel, counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
