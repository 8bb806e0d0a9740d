#!/bin/bash
# kernel trace of a short train bench -> GPU idle time inside a step (tools/trace_gaps.py)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o g -- python3 bench.py --steps 3 --warmup 1 --mode ${MODE:-train} --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS:-} > $OUT/bench.log 2>&1 < /dev/null
echo "rc=$?"
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$f" ${LAST:-} | tee $OUT/gaps.txt
find $OUT -name "*kernel_trace.csv" -delete
