#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag>
# Produces gpurun_out/<tag>/: bench JSON lines, rocprofv3 kernel stats, PMC passes, test log.
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -1 "$OUT/pytest_gpu.log"
python bench.py 2> "$OUT/bench_c3.err" | tail -1 > "$OUT/bench_c3.json"
python bench.py --config c2 --steps 50 --warmup 10 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c2.json"
python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c5.json"
python bench.py --colors precomp --steps 20 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c3_precomp_colors.json"
python bench.py --depth-gradient --steps 20 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c3_depth_gradient.json"
python tools/bench_views.py 2>/dev/null | tail -1 > "$OUT/bench_c4_views_1gpu.json"
python tools/bench_views.py --batch 16 2>/dev/null | tail -1 > "$OUT/bench_c4_views_1gpu_batch16.json"
python tools/bench_anchors.py 2>/dev/null | tail -1 > "$OUT/bench_anchors.json"
python tools/sweep_c5.py 2>/dev/null > "$OUT/sweep_c5.jsonl"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-secondary > "$OUT/bench_under_rocprof.log" 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace_anchors" -- python3 "$ROOT/tools/bench_anchors.py" --steps 10 > "$OUT/bench_anchors_under_rocprof.log" 2>&1 )
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc" > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.json"
python tools/pmc_traffic.py "gpurun_out/$TAG/pmc_summary.json" > "$OUT/pmc_traffic.json"
cut -c1-300 "$OUT/bench_c3.json"
