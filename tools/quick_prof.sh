#!/bin/bash
# per-kernel time of the bench (3 steps incl. warm-up), printed; raw output discarded
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_q
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 < /dev/null
timeout 60 python3 tools/quick_prof.py gpurun_out/prof_q/q_kernel_stats.csv 3 26 < /dev/null
rm -rf gpurun_out/prof_q
