#!/bin/bash
# round-5 A/B helper (GPU box): usage  bash tools/r5_ab.sh TAG "ENV=.. ENV2=.." ["ENV=.." ...]   -> gpurun_out/r5_ab_TAG.txt
# each arm: python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline (with the per-family HIP-event pass)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=$1; shift
out=gpurun_out/r5_ab_$tag.txt
: > $out
i=0
for arm in "$@"; do
  i=$((i+1))
  f=gpurun_out/r5_ab_${tag}_$i.json
  env $arm timeout 400 python bench.py --mode train --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline ${BENCH_EXTRA:-} > $f 2> gpurun_out/r5_ab_${tag}_$i.err
  python - "$arm" $f >> $out <<'PY'
import json, sys
arm, f = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f))
except Exception as e:
    print(arm, "FAILED", e); sys.exit(0)
print(f"== {arm}: {d['ms_per_step']:.2f} ms/step  {d['value']:.0f} tok/s  loss {d.get('final_loss')}  power {d.get('power')}")
for k, v in d.get("roofline", {}).get("families", {}).items():
    print(f"   {k:24s} {v['avg_launch_us']:8.1f} us x {v['launches']:4d}  share {v['share_of_step_time']:.4f}  {v['achieved']:7.1f} TF  {v['hbm_achieved_gbs']:7.0f} GB/s")
PY
done
cat $out
