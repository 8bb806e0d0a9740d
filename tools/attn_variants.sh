#!/bin/bash
# Build variants of csrc/attn_spatial.hip HERE into variants/ and time them on the GPU box (tools/attn_bench.py):
#   tools/attn_variants.sh build "name1:-DATTN_KT=1" "name2:-DATTN_KV_LDS" ...;   gpurun -- bash tools/attn_variants.sh run
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p variants
  rm -f variants/libhma_at_*.so
  OBJS=$(ls hma_amd/build/*.o | grep -v -E "/attn_spatial.o|/gemm_[a-z0-9_]+.o")
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $flags -c hma_amd/csrc/attn_spatial.hip -o variants/attn_$name.o 2>&1 | grep -E "error" ;
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libhma_at_$name.so $OBJS variants/attn_$name.o && rm variants/attn_$name.o ) &
  done
  wait
  ls variants/
else
  mkdir -p gpurun_out
  : > gpurun_out/attn_variants.txt
  for NTOK in 320 256; do
    echo "# n = $NTOK" | tee -a gpurun_out/attn_variants.txt
    (echo -n "default: "; NTOK=$NTOK timeout 120 python3 tools/attn_bench.py 2>&1 | tail -3 | tr '\n' ' '; echo) | tee -a gpurun_out/attn_variants.txt
    for so in variants/libhma_at_*.so; do
      (echo -n "$so: "; NTOK=$NTOK HMA_DEBUG_LIB=$so timeout 120 python3 tools/attn_bench.py 2>&1 | tail -3 | tr '\n' ' '; echo) | tee -a gpurun_out/attn_variants.txt
    done
  done
fi
