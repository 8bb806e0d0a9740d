#!/bin/bash
# ablation of the LDS-DMA ring NT kernel (debug build: python tools/phase_prof.py --build): 1 no A DMA, 2 no W DMA,
# 4 no epilogue traffic, 8 no MFMA
for a in ${1:-0 1 2 3 4 8 7 12 11 15}; do
  echo "== ablate $a"
  HMA_GEMM_ABLATE=$a HMA_DEBUG_LIB=hma_amd/libhma_hip_prof.so timeout 120 python3 tools/gemm_bench.py 2>&1 | grep "K1024\|K768" | grep -v wgrad | cut -c1-60
done
