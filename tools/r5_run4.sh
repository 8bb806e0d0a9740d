#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "gemm_tn" -p no:cacheprovider 2>&1 | tail -5
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_headline_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -5
bash tools/r5_ab.sh multi "HMA_WGRAD_MULTI=0" "HMA_WGRAD_MULTI=1" 2>&1 | grep -E "==|tn|wgrad"
} 2>&1 | tee gpurun_out/r5_run4.txt
