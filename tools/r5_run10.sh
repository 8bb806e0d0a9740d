#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_chain_gpu.py -q -x -p no:cacheprovider -k "chain_ab" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fulldepth_gpu.py -q -x -p no:cacheprovider -k "not decode" 2>&1 | tail -4
for lib in variants/libhma_ch_ssg.so hma_amd/libhma_hip.so variants/libhma_ch_ssg.so hma_amd/libhma_hip.so; do
  timeout 400 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --lib $lib > gpurun_out/r5_v.json 2> gpurun_out/r5_v.err
  python - $lib <<'PY'
import json,sys
d=json.load(open("gpurun_out/r5_v.json")); f=d["roofline"]["families"]
print(sys.argv[1], "%.2f ms"%d["ms_per_step"], " ".join("%s %.1f"%(k[4:],f[k]["avg_launch_us"]) for k in ("hma_chain_ab_fwd","hma_chain_a_bwd","hma_chain_s_bwd")), d["power"]["sclk_mhz"])
PY
done
} 2>&1 | tee gpurun_out/r5_run10.txt
