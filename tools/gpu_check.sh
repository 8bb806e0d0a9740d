#!/bin/bash
# usage: bash tools/gpu_check.sh [tests|bench|all]   (every step is bounded by `timeout`)
mkdir -p gpurun_out
what=${1:-all}
if [ "$what" = "tests" ] || [ "$what" = "all" ]; then
  timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x 2>&1 | tail -25 | tee gpurun_out/tests_latest.log
fi
if [ "$what" = "bench" ] || [ "$what" = "all" ]; then
  timeout 600 python bench.py --steps 5 --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -5 | tee gpurun_out/bench_latest.log
fi
