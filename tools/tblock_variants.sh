#!/bin/bash
# Build variants of the fused temporal-block probe HERE:  tools/tblock_variants.sh build "name:-DFLAGS" ...   and time them on the
# GPU box:  gpurun -- bash tools/tblock_variants.sh run   (results: gpurun_out/tblock_variants.txt)
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p variants
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-comment -Ihma_amd/csrc -include tools/probes/tblock_types.h $flags \
      tools/probes/tblock_fwd_experiment.hip -o variants/libtb_$name.so 2>&1 | grep -E "error" &
  done
  wait; ls variants/
else
  mkdir -p gpurun_out; : > gpurun_out/tblock_variants.txt
  for so in variants/libtb_*.so; do
    HMA_TB=$so timeout 120 python3 tools/tblock_bench.py 2>&1 | tail -1 | tee -a gpurun_out/tblock_variants.txt
  done
fi
