#!/bin/bash
# rocprofv3 kernel-trace stats of one bench mode (MODE=mar|decode|train): gpurun -- bash tools/prof_mode.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
MODE=${MODE:-mar}
OUT=gpurun_out/prof_$MODE
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $MODE -- python3 bench.py --steps 3 --warmup 1 --mode $MODE --no-cpu-baseline --no-kernel-timing > $OUT/bench.log 2>&1 < /dev/null
echo "trace rc=$?"
tail -1 $OUT/bench.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/summary.txt", "w") as o:
    for r in rows[:40]:
        line = f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):7d} calls {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/tot*100:6.2f}%  {r['Name'][:110]}"
        print(line); o.write(line + "\n")
    o.write(f"total {tot/1e6:.1f} ms\n")
PY
