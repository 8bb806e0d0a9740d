#!/bin/bash
# rocprofv3 kernel-trace stats of the STMAR bench (configs[3]) -> gpurun_out/mprof/stats.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mprof
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o q -- python3 bench.py --mode mar --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1 < /dev/null
echo "rc=$?"; tail -1 $OUT/bench.log | cut -c1-300
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%  {n[:90]}')
print(f"total {tot/1e6:.1f} ms over the traced steps")
PY
find $OUT -name "*kernel_trace.csv" -delete
