#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace by (kernel, grid, workgroup): calls and mean duration.  tools/trace_groups.py TRACE.csv [min_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    key = (name, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", ""))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot * 100 < lim:
        continue
    print(f"{100 * sum(v) / tot:5.2f}%  {len(v):5d} x {sum(v) / len(v):8.1f} us  grid {key[1]:>8s} x {key[2]:>3s}  {key[0][:70]}")
