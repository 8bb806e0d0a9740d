#!/bin/bash
# kernel-trace durations of the ablated ring kernel, one shape at a time:  tools/tn_ablate_trace.sh "0 8 7 15" "proj qkv_t"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export HMA_GEMM_TN_DMA=tr HMA_DEBUG_LIB=hma_amd/libhma_hip_prof.so
for sh in $2; do for a in $1; do
  rm -rf gpurun_out/prof_tn
  HMA_GEMM_TN_ABLATE=$a TN_SHAPES=$sh timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tn -o t -- python3 tools/tn_bench.py > /dev/null 2>&1
  echo "== $sh ablate $a"; python3 tools/trace_runs.py gpurun_out/prof_tn/t_kernel_trace.csv | cut -c1-40,60-130
done; done
rm -rf gpurun_out/prof_tn
