#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attn_spatial or headblocked or gemm_tn" -p no:cacheprovider 2>&1 | tail -6
timeout 300 python -m pytest tests/test_chain_gpu.py -q -k "chain_s" -p no:cacheprovider 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
bash tools/r5_ab.sh hb "HMA_ATTN_HB=0" "HMA_ATTN_HB=1" 2>&1 | tail -36
} 2>&1 | tee gpurun_out/r5_run2.txt
