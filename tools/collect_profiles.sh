#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag> [a|b|all]
#   part a: test log, bench lines, parity report, BloomScene-shaped trace;  part b: kernel stats, PMC passes, walk statistics
#   (two gpurun calls: together they exceed one call's 20-minute limit)
# Produces gpurun_out/<tag>/: bench JSON lines, rocprofv3 kernel stats, PMC passes (C3 and C5), walk statistics,
# parity report, test log.  Copy what is to be judged into profiles/<tag>/.
set -u
TAG=${1:-prof}
PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "$PART" != "b" ]; then
python -m pytest tests -m gpu -q -s > "$OUT/pytest_gpu.log" 2>&1; tail -1 "$OUT/pytest_gpu.log"
python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench_c3.err" | tail -1 > "$OUT/bench_c3_driver_form.json"   # what the driver runs
python bench.py 2>> "$OUT/bench_c3.err" | tail -1 > "$OUT/bench_c3.json"
python bench.py --exact-exp --steps 40 --warmup 10 --no-cpu-baseline --no-c4 --no-secondary 2>/dev/null | tail -1 > "$OUT/bench_c3_exact_exp.json"
python bench.py --config c2 --steps 100 --warmup 10 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c2.json"
python bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c5.json"
python bench.py --colors precomp --steps 50 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c3_precomp_colors.json"
python bench.py --depth-gradient --steps 50 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c3_depth_gradient.json"
python tools/bench_anchors.py 2>/dev/null | tail -1 > "$OUT/bench_anchors.json"
python tools/sweep_c5.py 2>/dev/null > "$OUT/sweep_c5.jsonl"
python tools/parity_report.py c1 c2 c3 c5 dense free_camera precomp lists > "$OUT/parity_report.jsonl" 2>/dev/null
# the BloomScene-shaped step: kernel trace -> idle intervals (tools/trace_gaps.py)
for m in default capacity graph; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt_bloomscene_shape_$m" -- python3 "$ROOT/tools/profile_bloomscene_shape.py" --steps 200 --mode $m > "$OUT/bloomscene_shape_${m}_under_rocprof.log" 2>&1 )
python tools/trace_gaps.py "$OUT/kt_bloomscene_shape_$m" > "$OUT/bloomscene_shape_gaps_$m.txt" 2>&1
python tools/profile_bloomscene_shape.py --steps 200 --mode $m 2>/dev/null | tail -1 > "$OUT/bench_bloomscene_shape_$m.json"
done
fi
if [ "$PART" != "a" ]; then
if [ -f bloomscene_amd/libbsr_rast_stats.so ]; then
  python tools/walk_stats.py --lib $ROOT/bloomscene_amd/libbsr_rast_stats.so 2>/dev/null | tail -1 > "$OUT/walk_stats_c3.json"
  python tools/walk_stats.py --lib $ROOT/bloomscene_amd/libbsr_rast_stats.so --scale-mul 3 2>/dev/null | tail -1 > "$OUT/walk_stats_c3_dense.json"
  python tools/walk_stats.py --lib $ROOT/bloomscene_amd/libbsr_rast_stats.so --config c5 2>/dev/null | tail -1 > "$OUT/walk_stats_c5.json"
fi
if [ -f bloomscene_amd/libbsr_rast_timeline.so ]; then
  python tools/walk_stats.py --timeline 2>/dev/null | tail -1 > "$OUT/timeline_c3.json"
  python tools/walk_stats.py --timeline --config c5 2>/dev/null | tail -1 > "$OUT/timeline_c5.json"
fi
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-secondary > "$OUT/bench_under_rocprof.log" 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace_c5" -- python3 "$ROOT/bench.py" --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-c4 > "$OUT/bench_c5_under_rocprof.log" 2>&1 )
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc" > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.json"
python tools/pmc_traffic.py "gpurun_out/$TAG/pmc_summary.json" > "$OUT/pmc_traffic.json"
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc_c5" --config c5 > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc_c5" > "$OUT/pmc_summary_c5.json"
for d in kernel_trace kernel_trace_c5; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d/kernel_trace/kernel_stats}.csv"; done
# keep the merged-back payload small: the raw per-dispatch CSVs stay on the box
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
fi
cut -c1-300 "$OUT/bench_c3_driver_form.json" 2>/dev/null
