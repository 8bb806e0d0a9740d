#!/bin/bash
# Where do the __amd_rocclr_copyBuffer dispatches of a bench run sit?  (kernel trace, neighbours of each copy)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ctrace
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o q -- python3 bench.py --steps 2 --warmup 1 --layers 4 --mode train --no-cpu-baseline --no-kernel-timing > $OUT/bench.log 2>&1 < /dev/null
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
prevc, nextc = collections.Counter(), collections.Counter()
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print("kernels", len(rows), "copies", len(idx))
for i in idx:
    prevc[names[i - 1] if i else "-"] += 1
    nextc[names[i + 1] if i + 1 < len(names) else "-"] += 1
print("before a copy:", prevc.most_common(12))
print("after a copy :", nextc.most_common(12))
# size of grids of copies
g = collections.Counter((r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?")) for i, r in enumerate(rows) if i in set(idx))
print("copy grid sizes:", g.most_common(8))
# position histogram: copies per 1000 kernels
h = collections.Counter(i // 500 for i in idx)
print("copies per 500-kernel window:", sorted(h.items()))
PY
rm -rf $OUT/trace
