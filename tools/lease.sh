#!/bin/bash
# One GPU lease = one case of this script:  gpurun --timeout N -- 'bash tools/lease.sh CASE [args]'  (output under gpurun_out/).
# bench_line LIB: one train-leg line of bench.py for library LIB, reduced to step time + the hot families.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
line() {  # line LIB [extra bench args]: prints "LIB ms/step family us ..."
  local lib=$1; shift
  timeout 500 python bench.py --mode train --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --lib "$lib" "$@" > gpurun_out/_line.json 2> gpurun_out/_line.err || { tail -5 gpurun_out/_line.err; return 1; }
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_line.json")); f = d["roofline"]["families"]
keys = ("hma_chain_ab_fwd", "wgrad_ring", "hma_mlp_bwd", "hma_chain_a_bwd", "hma_attn_spatial_bwd_blocked", "hma_chain_s_bwd", "hma_chain_t_bwd", "hma_attn_spatial_fwd")
print(sys.argv[1], "%.2f ms" % d["ms_per_step"], " ".join("%s %.1f" % (k.replace("hma_", ""), f[k]["avg_launch_us"]) for k in keys if k in f),
      d.get("power", {}).get("sclk_mhz"), "loss %.5f" % d.get("final_loss", float("nan")))
PY
}
case "$1" in
  tn_multi)  # weight-gradient launch: shipped library by problem set, then the debug build's ablations and phase timers
    {
      TN_SET="seven;six;mlp;attn;fc2;fc1;qkv_t;qkv_s;proj_t" timeout 300 python tools/tn_multi_bench.py
      for a in 0 1 2 4 6 8 16; do
        HMA_LIB=hma_amd/libhma_hip_prof.so HMA_GEMM_TN_ABLATE=$a timeout 120 python tools/tn_multi_bench.py
      done
      line hma_amd/libhma_hip.so
    } 2>&1 | tee gpurun_out/tn_multi.txt ;;
  tn_kinds)  # pure streaming rate (ablation 6: DMA + barriers only) and full rate per operand kind, same problem repeated
    {
      for set in "proj_t,proj_t,proj_t,proj_t,proj_t,proj_t,proj_t,proj_t" "qkv_t,qkv_t,qkv_t,qkv_t,qkv_t" "qkv_s,qkv_s,qkv_s,qkv_s,qkv_s" "fc1,fc1,fc1,fc1" "fc2,fc2,fc2,fc2" "seven"; do
        for a in 6 0; do
          TN_SET="$set" HMA_LIB=${LIB:-hma_amd/libhma_hip_prof.so} HMA_GEMM_TN_ABLATE=$a timeout 120 python tools/tn_multi_bench.py | grep -v "phase cycles"
        done
      done
    } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tn_kinds.txt ;;
  tn_libs)  # tn_libs LIB...: the seven-problem launch with each library (debug builds of gemm.hip variants), ablations 0 and 6
    shift
    {
      for lib in "$@"; do
        for a in 0 6; do
          echo "== $lib abl $a"
          TN_SET="seven" HMA_LIB=$lib HMA_GEMM_TN_ABLATE=$a timeout 120 python tools/tn_multi_bench.py
        done
      done
    } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tn_libs.txt ;;
  frames)  # frames LIB T...: the train leg at other window lengths (tokens/s = B T 256 / step)
    lib=$2; shift 2
    { for t in "$@"; do
        line $lib --frames $t
        python -c "import json; d=json.load(open('gpurun_out/_line.json')); print('  T=$t', '%.0f tokens/s' % d['value'], d['config'])"
      done; } 2>&1 | tee gpurun_out/frames.txt ;;
  prof)  # prof NAME bench-args...: rocprofv3 kernel-trace stats of one bench.py command -> gpurun_out/prof_NAME.txt (+ the csv)
    name=$2; shift 2
    out=gpurun_out/prof_$name
    rm -rf $out; mkdir -p $out
    ( cd /tmp && true )
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o q -- python3 bench.py "$@" > $out/bench.log 2>&1 < /dev/null
    echo "rc=$?"; tail -1 $out/bench.log | cut -c1-400
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    cp "$f" gpurun_out/prof_$name.csv
    python3 - "$f" <<'PY' | tee gpurun_out/prof_$name.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%  {n[:100]}')
print(f"total {tot/1e6:.1f} ms")
PY
    find $out -name "*kernel_trace.csv" -delete ;;
  dpcheck)  # the bench line's train_T12 and dp_check sub-objects alone (child processes of bench.py)
    python -c "import json, bench; print(json.dumps({'train_T12': bench.child_line(['--frames', '12'], {}), 'dp_check': bench.forced_collectives_check()}, indent=1))" 2>&1 | tee gpurun_out/dpcheck.txt ;;
  mode_ab)  # mode_ab MODE LIB_A LIB_B: interleaved same-box A / B of one bench leg (decode | mar) for two libraries, twice each
    m=$2; a=$3; b=$4
    { for lib in $a $b $a $b; do
        timeout 600 python bench.py --mode $m --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline --lib $lib > gpurun_out/_mode.json 2> gpurun_out/_mode.err || tail -5 gpurun_out/_mode.err
        python -c "import json; d=json.load(open('gpurun_out/_mode.json')); print('$m $lib %.1f %s %.2f ms/step' % (d['value'], d['unit'], d.get('ms_per_step', float('nan'))), d.get('latency_b1', ''))"
      done; } 2>&1 | tee gpurun_out/mode_ab_$m.txt ;;
  mode)  # mode MODE [env assignments...]: one bench leg (decode | mar), its line reduced to the value and the hot kernels
    m=$2; shift 2
    { for e in "$@" ""; do
        env $e timeout 600 python bench.py --mode $m --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline > gpurun_out/_mode.json 2> gpurun_out/_mode.err || tail -5 gpurun_out/_mode.err
        python - "$m" "$e" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_mode.json"))
print(sys.argv[1], sys.argv[2] or "(default)", "%.1f %s" % (d["value"], d["unit"]), "%.2f ms/step" % d.get("ms_per_step", float("nan")), d.get("latency_b1", ""),
      ("host issue %.1f ms/step" % d["host_issue_ms_per_step"]) if "host_issue_ms_per_step" in d else "")
PY
      done; } 2>&1 | tee gpurun_out/mode_$m.txt ;;
  ab)  # ab LIB_A LIB_B [bench args]: interleaved same-box A / B of two libraries, twice each
    a=$2; b=$3; shift 3
    { for lib in $a $b $a $b; do line $lib "$@"; done; } 2>&1 | tee gpurun_out/ab.txt ;;
  tests)  # the GPU suite (optionally -k EXPR) + smoke
    shift
    { timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider "$@" 2>&1 | tail -15
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3; } | tee gpurun_out/tests.txt ;;
  *) echo "unknown case $1"; exit 2 ;;
esac
