#!/bin/bash
# Time the default build and every variants/libhma_ch_*.so with tools/chain_bench.py at the train (M = 163840) and decode-frame (M = 20480) sizes.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/chain_ab.txt
: > $out
for M in ${MS:-163840 20480}; do
  echo "# M = $M" | tee -a $out
  M=$M timeout 200 python3 tools/chain_bench.py 2>&1 | tail -2 | tee -a $out
  for so in variants/libhma_ch_*.so; do
    M=$M HMA_LIB=$so timeout 200 python3 tools/chain_bench.py 2>&1 | tail -2 | tee -a $out
  done
done
