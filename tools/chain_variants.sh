#!/bin/bash
# Build variants of csrc/chain.hip HERE (hipcc cross-compiles) into variants/ (git-ignored, travels with gpurun):
#   tools/chain_variants.sh build "name1:-DCH_ABL=1" "name2:-DFOO" ...
# and time them on the GPU box:  gpurun -- bash tools/chain_variants.sh run
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p variants
  rm -f variants/libhma_ch_*.so
  OBJS=$(ls hma_amd/build/*.o | grep -v -E "/chain.o|/gemm_[a-z0-9_]+.o")
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $flags -c hma_amd/csrc/chain.hip -o variants/chain_$name.o 2>&1 | grep -E "error" ;
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libhma_ch_$name.so $OBJS variants/chain_$name.o && rm variants/chain_$name.o ) &
    while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
  done
  wait
  ls variants/
else
  mkdir -p gpurun_out
  : > gpurun_out/chain_variants.txt
  timeout 120 python3 tools/chain_bench.py 2>&1 | grep "chain [ABS]" | tee -a gpurun_out/chain_variants.txt
  for so in variants/libhma_ch_*.so; do
    HMA_LIB=$so timeout 120 python3 tools/chain_bench.py 2>&1 | grep "chain [ABS]" | tee -a gpurun_out/chain_variants.txt
  done
fi
