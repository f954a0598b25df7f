#!/bin/bash
# Where does the bucket form of the second binning pass stop paying?  Step time of the product library (host-side choice,
# binning.hip: binning_plan) against the two pinned builds on frames of growing density (C3's scene, scales multiplied):
#   bash tools/plan_sweep.sh "1.0 1.5 2.0 2.5 3.0"        (one GPU box; make debug_variants first)
#   bash tools/plan_sweep.sh "1.0" "--config c5"          (extra bench arguments)
for sm in ${1:-1.0 1.5 2.0 2.5 3.0}; do
  for L in libbloomscene_rast.so libbsr_chain_only.so libbsr_bucket_always.so libbsr_bucket_big.so; do
    python bench.py --lib $PWD/bloomscene_amd/$L --no-cpu-baseline --no-c4 --no-secondary --steps 30 --warmup 8 --scale-mul $sm ${2:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d['stage_ms']
print('scale %s %-26s %8.1f Msplats/s %7.4f ms  R/tile %6.0f  binning %.4f sort_tiles %.4f' % ('$sm', '$L', d['value'], d['ms_per_step'], d['config'].get('num_rendered', 0) / 8160.0, s.get('binning', 0), s.get('sort_tiles', 0)) + '  median %.4f max %.4f' % (d['config']['ms_per_step_median'], d['config']['step_ms_max']))"
  done
done
