#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for lib in hma_amd/libhma_hip.so variants/libhma_ch_noss.so variants/libhma_ch_nost.so; do
  timeout 400 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --lib $lib > gpurun_out/r5_v.json 2> gpurun_out/r5_v.err
  python - $lib <<'PY'
import json,sys
d=json.load(open("gpurun_out/r5_v.json")); f=d["roofline"]["families"]
print(sys.argv[1], "%.2f ms"%d["ms_per_step"], " ".join("%s %.1f"%(k[4:],f[k]["avg_launch_us"]) for k in ("hma_chain_ab_fwd","hma_chain_a_bwd","hma_chain_s_bwd")), d["power"]["sclk_mhz"])
PY
done
timeout 2300 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout 1200 2>&1 | tail -6
} 2>&1 | tee gpurun_out/r5_run9.txt
