#!/bin/bash
# SQ counters per kernel of an 8-layer bench step (separate pass: no trace domains), summary to gpurun_out/pmc_sq_$ROUND.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r1}
rm -rf gpurun_out/pmc_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d gpurun_out/pmc_sq -o q -- python3 bench.py --steps 1 --warmup 1 --layers 8 --mode train --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 < /dev/null
timeout 120 python3 tools/pmc_sq.py gpurun_out/pmc_sq > gpurun_out/pmc_sq_$R.txt < /dev/null
rm -rf gpurun_out/pmc_sq
head -40 gpurun_out/pmc_sq_$R.txt
