cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_bench.log 2>&1
ls -R gpurun_out/prof | head -30
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
echo $f; head -40 $f
python __graft_entry__.py smoke 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -15
