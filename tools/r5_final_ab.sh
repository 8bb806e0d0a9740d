#!/bin/bash
# same-box A/B of the whole round: the round-4 launch sequence (switches off, 7-wave attention backward, whole tiles dealt round-robin;
# the forked weight gradients, worth nothing, are gone in both; variants/libhma_r4like.so = this tree with -DCH_RAGGED=0 (chain.hip) and
# -DATTN_BAL=0 (attn_spatial.hip: the round-4 attention backward, which keeps this round's scratch-free staging)) against this round's default
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r5_final_ab.txt
for rep in 1 2; do
  HMA_CHAIN_S=0 HMA_ATTN_HB=0 HMA_WGRAD_MULTI=0 HMA_CHAIN_AB=0 HMA_CHAIN_T=0 timeout 400 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --lib variants/libhma_r4like.so > gpurun_out/r5_fab.json 2> gpurun_out/r5_fab.err
  python -c "import json;d=json.load(open('gpurun_out/r5_fab.json'));print('round-4 launch sequence: %.2f ms/step  %.0f tok/s'%(d['ms_per_step'],d['value']))" | tee -a gpurun_out/r5_final_ab.txt
  timeout 400 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/r5_fab.json 2> gpurun_out/r5_fab.err
  python -c "import json;d=json.load(open('gpurun_out/r5_fab.json'));print('round 5 default:         %.2f ms/step  %.0f tok/s'%(d['ms_per_step'],d['value']))" | tee -a gpurun_out/r5_final_ab.txt
done
