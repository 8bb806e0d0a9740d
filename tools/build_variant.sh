#!/bin/bash
# build a variant of the library with extra -D flags for gemm.hip:  tools/build_variant.sh NAME -DFOO=1 ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c hma_amd/csrc/gemm.hip -o hma_amd/build/gemm_$name.o
objs=$(ls hma_amd/build/*.o | grep -v "gemm" )
hipcc --offload-arch=gfx950 -shared -fPIC -o hma_amd/libhma_hip_$name.so hma_amd/build/gemm_$name.o $objs
echo hma_amd/libhma_hip_$name.so
