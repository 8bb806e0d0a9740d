cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for w in ln gemm attn; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/probe_$w -o p -- python3 tools/copy_probe.py $w > /dev/null 2>&1 < /dev/null
  echo "== $w"; cut -d, -f1,2 gpurun_out/probe_$w/p_kernel_stats.csv | head -6
done
