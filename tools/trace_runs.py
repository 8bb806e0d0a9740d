#!/usr/bin/env python3
"""Mean duration of each consecutive run of identical (kernel, grid) dispatches in a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    else:
        runs.append((key, [d]))
# interleaved kernels (main + reduce) break runs: merge alternating patterns by key order of appearance
agg = {}
order = []
seg = 0
last_shape = None
for key, ds in runs:
    if key not in agg:
        agg[key] = []
        order.append(key)
    agg[key].extend(ds)
for key in order:
    ds = agg[key]
    if len(ds) < 3 or "tn" not in key[0]:
        continue
    ds2 = ds[3:] if len(ds) > 6 else ds
    print(f"{key[0][:60]:60s} grid {key[1]:>8s} x {key[2]:>4s}  n={len(ds):3d}  mean {sum(ds2) / len(ds2):8.1f} us  min {min(ds2):8.1f}")
