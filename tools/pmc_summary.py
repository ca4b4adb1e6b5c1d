"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (sum over dispatches)."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
print(files)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        k = name[:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, val in sorted(v.items()):
        print(f"    {c:28s} {val / max(1, cnt[(k, c)]):16.1f} per dispatch ({cnt[(k, c)]} dispatches)")
