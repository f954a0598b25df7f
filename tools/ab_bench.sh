#!/bin/bash
# A/B builds of the library on ONE GPU box (boxes differ by +-4 %): tools/ab_bench.sh [-r ROUNDS] [-a "bench args"] libA.so libB.so ...
# Interleaved runs; prints the bench JSON's value / ms_per_step / stage_ms per run.
# One call = one box.  Boxes differ by +-4 % and can disagree on the SIGN of a 2 % effect (docs/EXPERIMENTS.md, round 4:
# pair-padded lists won on one box and lost on six): confirm a small difference on two or more boxes before adopting it.
ROUNDS=3; ARGS=""
while getopts "r:a:" o; do case $o in r) ROUNDS=$OPTARG;; a) ARGS=$OPTARG;; esac; done
shift $((OPTIND-1))
for i in $(seq $ROUNDS); do
  for L in "$@"; do
    python bench.py --lib $PWD/$L --no-cpu-baseline --no-c4 --no-secondary --steps 40 --warmup 10 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-40s %8.1f %7.4f  ' % ('$L', d['value'], d['ms_per_step']) + ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items()))"
  done
done
