#!/bin/bash
# rocprofv3 kernel-trace stats + two PMC passes (HBM read / write bytes) of one bench leg; every step time-bounded.
#   ROUND=r4 MODE=train|decode|mar  gpurun -- bash tools/prof_bench.sh
# Writes gpurun_out/prof_<round>_<mode>/{kernel_stats_<round>.csv, pmc_<round>.json}; the PMC passes run in their own processes with
# --pmc only (no trace domains).  train: 8-layer model (bytes per launch do not depend on the depth); mar: the full depth, three steps;
# decode: a 4-layer model, two rollouts (a full-depth rollout is ~14 000 launches: too slow under --pmc), scaled by 32 / 4 -- the
# embedding / readout / sampling kernels outside the layers are < 2 % of a rollout's bytes.  `summary` = bytes of the measured unit.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r1}
MODE=${MODE:-train}
OUT=gpurun_out/prof_${R}_$MODE
rm -rf $OUT; mkdir -p $OUT
case $MODE in
  # (train: the train leg of the driver's command -- `bench.py --steps 20 --warmup 5` -- WITH its HIP-event pass, so that the printed line and
  # the kernel summary come from one process: tools/roofline_check.py compares them)
  train)  TR="--steps 20 --warmup 5 --mode train --no-cpu-baseline"; PM="--steps 1 --warmup 1 --layers 8 --mode train --no-cpu-baseline --no-kernel-timing";;
  decode) TR="--steps 2 --warmup 2 --mode decode --no-cpu-baseline --no-latency"; PM="--steps 1 --warmup 1 --layers 4 --mode decode --no-cpu-baseline --no-latency";;
  mar)    TR="--steps 3 --warmup 2 --mode mar --no-cpu-baseline"; PM="--steps 1 --warmup 2 --mode mar --no-cpu-baseline";;
esac
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $R -- python3 bench.py $TR > $OUT/bench_trace.log 2>&1 < /dev/null
echo "trace rc=$?"
grep -h '^{"metric"' $OUT/bench_trace.log | tail -1 > $OUT/bench_line_$R.json
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $R -- python3 bench.py $PM > $OUT/bench_pmc_fetch.log 2>&1 < /dev/null
echo "pmc fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $R -- python3 bench.py $PM > $OUT/bench_pmc_write.log 2>&1 < /dev/null
echo "pmc write rc=$?"
timeout 120 python3 tools/prof_summary.py $OUT $R $MODE < /dev/null
# raw per-dispatch CSVs are large: keep only the summaries + stats
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
