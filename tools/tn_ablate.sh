#!/bin/bash
# ablation of the LDS-DMA ring wgrad kernel (debug build: python tools/phase_prof.py --build); TN_SHAPES as tools/tn_bench.py
# bits: 1 no DMA, 2 no MFMA, 4 no LDS reads, 8 no partial stores, 16 no barrier
for a in ${ABL:-0 8 1 9 3 7 15}; do
  echo "== ablate $a"
  HMA_GEMM_TN_ABLATE=$a HMA_GEMM_TN_DMA=tr HMA_DEBUG_LIB=hma_amd/libhma_hip_prof.so timeout 120 python3 tools/tn_bench.py 2>&1 | grep wgrad
done
