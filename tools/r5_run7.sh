#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_chain_gpu.py -q -x -p no:cacheprovider -k "chain_ab" 2>&1 | tail -25 | tee gpurun_out/r5_run7.txt
