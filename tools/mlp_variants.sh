#!/bin/bash
# Build variants of csrc/mlp.hip HERE (build container, hipcc cross-compiles) into variants/ (git-ignored, travels with gpurun):
#   tools/mlp_variants.sh build "name1:-DFOO -DBAR=1" "name2:-DMLP_ABL=4" ...
# and time them on the GPU box:
#   gpurun -- bash tools/mlp_variants.sh run
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p variants
  OBJS=$(ls hma_amd/build/*.o | grep -v -E "/mlp.o")
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $flags -c hma_amd/csrc/mlp.hip -o variants/mlp_$name.o 2>&1 | grep -E "error" ;
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libhma_$name.so $OBJS variants/mlp_$name.o && rm variants/mlp_$name.o ) &
    while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
  done
  wait
  ls variants/
else
  mkdir -p gpurun_out
  : > gpurun_out/mlp_variants.txt
  timeout 120 python3 tools/mlp_bench.py 2>&1 | tail -1 | tee -a gpurun_out/mlp_variants.txt
  for so in variants/libhma_*.so; do
    case $so in
      *prof*) HMA_LIB=$so MLP_PROF=1 timeout 120 python3 tools/mlp_bench.py 2>&1 | tail -21 | tee -a gpurun_out/mlp_variants.txt ;;
      *) HMA_LIB=$so timeout 120 python3 tools/mlp_bench.py 2>&1 | tail -1 | tee -a gpurun_out/mlp_variants.txt ;;
    esac
  done
fi
