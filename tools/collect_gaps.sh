#!/bin/bash
# The BloomScene-shaped step in its three forms under rocprofv3 --kernel-trace -> idle intervals (tools/trace_gaps.py).
# usage (GPU box, repo root): bash tools/collect_gaps.sh <tag>
TAG=${1:-prof}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for m in default capacity graph; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt_bloomscene_shape_$m" -- python3 "$ROOT/tools/profile_bloomscene_shape.py" --steps 200 --mode $m > "$OUT/bloomscene_shape_${m}_under_rocprof.log" 2>&1 )
python tools/trace_gaps.py "$OUT/kt_bloomscene_shape_$m" > "$OUT/bloomscene_shape_gaps_$m.txt" 2>&1
python tools/profile_bloomscene_shape.py --steps 200 --mode $m 2>/dev/null | tail -1 > "$OUT/bench_bloomscene_shape_$m.json"
find "$OUT/kt_bloomscene_shape_$m" -name "*.csv" -delete
head -1 "$OUT/bloomscene_shape_gaps_$m.txt"
done
