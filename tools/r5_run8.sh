#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_headline_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_fulldepth_gpu.py -q -x -p no:cacheprovider -k "forward_backward or trainer_graph" 2>&1 | tail -4
bash tools/r5_ab.sh ab "HMA_CHAIN_AB=0" "HMA_CHAIN_AB=1" "HMA_CHAIN_AB=0" "HMA_CHAIN_AB=1" 2>&1 | grep -E "==|chain_a_fwd|chain_b_fwd|chain_ab|temporal_fwd"
} 2>&1 | tee gpurun_out/r5_run8.txt
