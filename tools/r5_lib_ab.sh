#!/bin/bash
# same-box A / B of library builds (GPU box): usage  bash tools/r5_lib_ab.sh TAG LIB1 LIB2 ...   ("default" = hma_amd/libhma_hip.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=$1; shift
out=gpurun_out/r5_lib_ab_$tag.txt
: > $out
i=0
for lib in "$@"; do
  i=$((i+1))
  f=gpurun_out/r5_lib_ab_${tag}_$i.json
  extra=""; [ "$lib" != default ] && extra="--lib $lib"
  timeout 400 python bench.py --mode train --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline $extra > $f 2> gpurun_out/r5_lib_ab_${tag}_$i.err
  python - "$lib" $f >> $out <<'PY'
import json, sys
arm, f = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f))
except Exception as e:
    print(arm, "FAILED", e); sys.exit(0)
print(f"== {arm}: {d['ms_per_step']:.2f} ms/step  {d['value']:.0f} tok/s  power {d.get('power', {}).get('sclk_mhz')} MHz")
for k, v in d.get("roofline", {}).get("families", {}).items():
    print(f"   {k:24s} {v['avg_launch_us']:8.1f} us x {v['launches']:4d}  share {v['share_of_step_time']:.4f}")
PY
done
cat $out
