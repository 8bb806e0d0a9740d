#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attn_spatial" -p no:cacheprovider 2>&1 | tail -4
timeout 400 python -m pytest tests/test_headline_gpu.py -q -k "attn_spatial" -p no:cacheprovider 2>&1 | tail -3
bash tools/attn_variants.sh run 2>&1 | tail -8
bash tools/r5_ab.sh attnbal "HMA_X=0" 2>&1 | tail -16
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -k "trainer_graph" -p no:cacheprovider 2>&1 | tail -5
timeout 900 python -m pytest tests/test_fulldepth_stmar_gpu.py -q -k "b4_within" -p no:cacheprovider 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r5_run1.txt
