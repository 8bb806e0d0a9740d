#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel name: tools/pmc_sum.py DIR [substring]"""
import csv, glob, os, sys, collections
files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        key = (name, r["Counter_Name"], r.get("Grid_Size", ""))
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
for (name, ctr, grid), (n, v) in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(f"{name[:50]:50s} grid {grid:>8s} {ctr:12s} launches {n:4d}  per launch {v / n:14.1f}")
