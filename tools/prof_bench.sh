#!/bin/bash
# rocprofv3 kernel-trace stats + two PMC passes (HBM read / write bytes) of the bench; every step time-bounded.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r1}
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $R -- python3 bench.py --steps 3 --warmup 1 --mode train --no-cpu-baseline --no-kernel-timing > $OUT/bench_trace.log 2>&1 < /dev/null
echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $R -- python3 bench.py --steps 1 --warmup 1 --layers 8 --mode train --no-cpu-baseline --no-kernel-timing > $OUT/bench_pmc_fetch.log 2>&1 < /dev/null
echo "pmc fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $R -- python3 bench.py --steps 1 --warmup 1 --layers 8 --mode train --no-cpu-baseline --no-kernel-timing > $OUT/bench_pmc_write.log 2>&1 < /dev/null
echo "pmc write rc=$?"
find $OUT -name "*.csv" -size +0 | head -20
timeout 120 python3 tools/prof_summary.py $OUT $R < /dev/null
# raw per-dispatch CSVs are large: keep only the summaries + stats
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
