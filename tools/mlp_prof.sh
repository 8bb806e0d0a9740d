#!/bin/bash
# Per-phase cycle counts of the fused MLP forward kernel (debug build with -DMLP_PROF), block 0, per wave.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OBJS=$(ls hma_amd/build/*.o | grep -v -E "mlp.o|gemm_prof|gemm_s4")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DMLP_PROF ${MLP_DEFS:-} -c hma_amd/csrc/mlp.hip -o /tmp/mlp_prof.o 2>&1 | grep error
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libhma_prof.so $OBJS /tmp/mlp_prof.o
HMA_LIB=/tmp/libhma_prof.so MLP_PROF=1 timeout 120 python3 tools/mlp_bench.py 2>&1 | tail -14 | tee gpurun_out/mlp_prof.txt
