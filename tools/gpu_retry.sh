#!/bin/bash
# usage: tools/gpu_retry.sh TIMEOUT LOGFILE -- command...   (retries while gpurun reports "busy": exit code 3)
t="$1"; log="$2"; shift 3
for i in $(seq 1 40); do
  gpurun --timeout "$t" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" "$log"; then exit $rc; fi
  sleep 120
done
exit 3  # every attempt was refused as busy: the command never ran
