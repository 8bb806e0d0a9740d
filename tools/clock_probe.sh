#!/bin/bash
# shader / memory clocks and power while one wgrad shape loops (debug build ablations): is the "additive" DMA + compute time a clock effect?
for a in 0 6 1 2; do
  echo "== ablate $a"
  rm -f /tmp/cp_$a.log
  HMA_GEMM_TN_ABLATE=$a HMA_GEMM_TN_DMA=tr HMA_DEBUG_LIB=hma_amd/libhma_hip_prof.so TN_SHAPES=fc2 TN_REPS=100000 timeout 200 python3 -u tools/tn_bench.py > /tmp/cp_$a.log 2>&1 &
  pid=$!
  for i in $(seq 1 600); do grep -q START /tmp/cp_$a.log 2>/dev/null && break; sleep 0.25; done
  sleep 1.5
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power (W)" | tr -s '\t ' ' ' | tr '\n' ';'; echo
    sleep 0.5
  done
  wait $pid
  grep wgrad /tmp/cp_$a.log
done
