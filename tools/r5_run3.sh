#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "blocked or headblocked" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python -m pytest tests/test_chain_gpu.py -q -k "chain_s" -p no:cacheprovider 2>&1 | tail -3
NTOK=320 timeout 120 python3 tools/attn_bench.py 2>&1 | tail -3
timeout 200 python3 tools/chain_bench.py 2>&1 | grep "chain S"
bash tools/r5_ab.sh hb "HMA_ATTN_HB=0" "HMA_ATTN_HB=1" 2>&1 | grep -E "==|attn_spatial_bwd|chain_s|tn_pair"
} 2>&1 | tee gpurun_out/r5_run3.txt
