#!/bin/bash
# Ablation of the fused MLP kernels: rebuilds csrc/mlp.hip with -DMLP_ABL=<bits> into /tmp and times each variant.
# bits: 1 no LDS-DMA, 2 no MFMA, 4 no GELU, 8 no barrier
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OBJS=$(ls hma_amd/build/*.o | grep -v -E "mlp.o|gemm_prof|gemm_s4")
: > gpurun_out/mlp_ablate.txt
timeout 120 python3 tools/mlp_bench.py 2>&1 | tail -1 | tee -a gpurun_out/mlp_ablate.txt
for abl in ${ABLS:-1 2 4 6 7 8}; do
  # (a non-numeric entry is passed as a macro name: e.g. MLP_NOROT)
  if [[ "$abl" =~ ^[0-9]+$ ]]; then def="-DMLP_ABL=$abl"; else def="-D$abl"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $def -c hma_amd/csrc/mlp.hip -o /tmp/mlp_$abl.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libhma_abl$abl.so $OBJS /tmp/mlp_$abl.o
  HMA_LIB=/tmp/libhma_abl$abl.so timeout 120 python3 tools/mlp_bench.py 2>&1 | tail -1 | tee -a gpurun_out/mlp_ablate.txt
done
