#!/bin/bash
# shader clock and package power about once a second while a bench leg runs (MODE=train|decode|mar, STEPS): is the step power-capped?
MODE=${MODE:-train}
python3 bench.py --mode $MODE --steps ${STEPS:-150} --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > /tmp/pt_$MODE.log 2>&1 &
pid=$!
SECONDS=0
while kill -0 $pid 2>/dev/null; do
  s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/.*: *//' | tr '\n' ' ')
  echo "${SECONDS} s  $s"
  sleep 0.7
done
tail -1 /tmp/pt_$MODE.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); d = d.get('$MODE', d) if '$MODE' != 'train' else d
print('$MODE', d['value'], d['unit'], d['ms_per_step'], 'ms per step')"
