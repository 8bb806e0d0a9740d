#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 2300 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 1200 --deselect tests/test_fulldepth_gpu.py --deselect tests/test_fulldepth_stmar_gpu.py 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_fulldepth_gpu.py tests/test_fulldepth_stmar_gpu.py -q -x -p no:cacheprovider --timeout 1200 2>&1 | tail -4
for lib in hma_amd/libhma_hip_bytes.so hma_amd/libhma_hip.so hma_amd/libhma_hip_bytes.so hma_amd/libhma_hip.so; do
  timeout 400 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --lib $lib > gpurun_out/r5_split.json 2> gpurun_out/r5_split.err
  python - $lib <<'PY'
import json,sys
d=json.load(open("gpurun_out/r5_split.json")); f=d["roofline"]["families"]
print(sys.argv[1], "%.2f ms"%d["ms_per_step"], "multi %.1f us"%f["hma_gemm_tn_multi"]["avg_launch_us"], d["power"]["sclk_mhz"])
PY
done
} 2>&1 | tee gpurun_out/r5_run5.txt
