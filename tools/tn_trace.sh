#!/bin/bash
# kernel-trace of tools/tn_pair_bench.py: durations of the ring kernel and of the reduction per shape (HMA_LIB, TN_SHAPES as there)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tn_trace
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/tn_pair_bench.py > $OUT/bench.log 2>&1 < /dev/null
echo "rc=$?"; grep -v amdgpu.ids $OUT/bench.log
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tn_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a shape = 30 calls (12 warm + 18 timed); the last 18 of each run of 30 are averaged
seq = [("dma" if "dma" in r["Kernel_Name"] else "red", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
per = 60  # dma + red per call
for s in range(0, len(seq), per):
    blk = seq[s + 24:s + per]
    d = [u for k, u in blk if k == "dma"]; r = [u for k, u in blk if k == "red"]
    if d: print(f"shape {s // per}: ring kernel {sum(d) / len(d):7.1f} us   reduction {sum(r) / max(len(r), 1):6.1f} us   ({len(d)} / {len(r)} launches)")
PY
find $OUT -name "*kernel_trace.csv" -delete
