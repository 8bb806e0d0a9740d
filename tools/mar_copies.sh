#!/bin/bash
# Where do the __amd_rocclr_copyBuffer dispatches of the STMAR bench sit (model construction, or every step)?
# gpurun -- bash tools/mar_copies.sh   -> gpurun_out/mar_copies.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mcopies
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o q -- python3 bench.py --steps 3 --warmup 2 --layers 4 --mode mar --no-cpu-baseline > $OUT/bench.log 2>&1 < /dev/null
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/mar_copies.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print("bench.py --mode mar --layers 4 --steps 3 --warmup 2: kernels", len(rows), "copyBuffer dispatches", len(idx))
# the optimizer's AdamW launches mark the end of a step
ends = [i for i, n in enumerate(names) if n.startswith("adamw")]
steps, last = [], -1
for i, e in enumerate(ends):
    if i + 1 == len(ends) or ends[i + 1] - e > 50:
        steps.append((last + 1, e))
        last = e
print("step boundaries (kernel index ranges):", steps)
for k, (a, b) in enumerate(steps):
    c = [i for i in idx if a <= i <= b]
    t = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in c) / 1e3
    tot = (int(rows[b]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    print(f"  range {k}: {b - a + 1:6d} kernels, {len(c):5d} copies ({t:8.1f} us of {tot:10.1f} us)")
    if k >= 1:
        prevc = collections.Counter(names[i - 1] for i in c)
        nextc = collections.Counter(names[i + 1] for i in c if i + 1 < len(names))
        print("     before a copy:", prevc.most_common(6))
        print("     after a copy :", nextc.most_common(6))
PY
rm -rf $OUT
