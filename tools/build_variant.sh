#!/bin/bash
# build a variant of the library with extra -D flags for ONE source file:  tools/build_variant.sh NAME [SRC.hip] -DFOO=1 ...
# (SRC defaults to gemm.hip; the other objects come from hma_amd/build/ -- run `python -m hma_amd.build` first)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=gemm
case "$1" in *.hip) src=${1%.hip}; shift ;; esac
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c hma_amd/csrc/$src.hip -o hma_amd/build/_var_$name.o
objs=$(ls hma_amd/build/*.o | grep -v "/_var_\|/${src}\.o\|_prof\.o" )
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libhma_$name.so hma_amd/build/_var_$name.o $objs
echo variants/libhma_$name.so
