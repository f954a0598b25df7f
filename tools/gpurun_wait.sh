#!/bin/bash
# gpurun, retried ONLY while the pod has no free slot/box (exit 3: nothing ran, nothing was charged).
# usage: tools/gpurun_wait.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
