"""Aggregate a rocprofv3 --pmc run: per kernel, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if pat and not any(p_ in k for p_ in pat.split(",")):
            continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v) / len(v):.4g}")
