#!/bin/bash
# Kernel-trace averages of the per-tile sort kernels (and the step) for one or more builds, on the dense leg, C5 and C3:
#   bash tools/sort_ab.sh <tag> product [libX.so ...]      (one GPU box; output under gpurun_out/<tag>*)
TAG=$1; shift
for cfg in "dense:--steps 20 --warmup 5 --scale-mul 3" "c5:--config c5 --steps 10 --warmup 3" "c3:--steps 20 --warmup 5"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 400 bash tools/kstats_ab.sh ${TAG}_$name "$args" "$@" > gpurun_out/${TAG}_$name.txt 2>&1
  echo "-- $name"; grep -E "==|sort_tiles|bucket_sort" gpurun_out/${TAG}_$name.txt
done
