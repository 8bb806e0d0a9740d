cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# train: kernel trace only (the byte counters of profiles/pmc_hbm_r4.json do not change with the late kernel changes of the round)
OUT=gpurun_out/prof_r4_train2; rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r4 -- python3 bench.py --steps 3 --warmup 1 --mode train --no-cpu-baseline --no-kernel-timing > $OUT/bench_trace.log 2>&1 < /dev/null
timeout 120 python3 tools/prof_summary.py $OUT r4 train < /dev/null | head -14
find $OUT -name "*kernel_trace.csv" -delete
