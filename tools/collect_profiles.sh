#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag> [a|b|all]
#   part a: test log, bench lines (driver form, full, C2, C5, precomp, depth gradient), parity report (default | strict | floor);
#   part b: kernel stats (C3, dense leg, C5), PMC passes + traffic (C3, dense leg)
#   (two gpurun calls: together they exceed one call's 20-minute limit)
# Produces gpurun_out/<tag>/...  Copy what is to be judged into profiles/<tag>/ (one mid-round and one final collection per round).
set -u
TAG=${1:-prof}
PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "$PART" != "b" ]; then
python -m pytest tests -m gpu -q -s > "$OUT/pytest_gpu.log" 2>&1; tail -1 "$OUT/pytest_gpu.log"
# what the driver runs: the LAST stdout line is the compact contract line, the detail record lies beside it
python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_c3_driver_form_detail.json" 2> "$OUT/bench_c3.err" | tail -1 > "$OUT/bench_c3_driver_form.json"
python bench.py --full --detail "$OUT/bench_c3_full_detail.json" 2>> "$OUT/bench_c3.err" | tail -1 > "$OUT/bench_c3_full.json"
python bench.py --config c2 --steps 100 --warmup 10 --no-cpu-baseline --no-c4 --detail "$OUT/bench_c2_detail.json" 2>/dev/null | tail -1 > "$OUT/bench_c2.json"
python bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline --no-c4 --detail "$OUT/bench_c5_detail.json" 2>/dev/null | tail -1 > "$OUT/bench_c5.json"
python bench.py --colors precomp --steps 50 --warmup 5 --no-cpu-baseline --no-c4 --detail "$OUT/bench_c3_precomp_detail.json" 2>/dev/null | tail -1 > "$OUT/bench_c3_precomp_colors.json"
python bench.py --depth-gradient --steps 50 --warmup 5 --no-cpu-baseline --no-c4 --detail "$OUT/bench_c3_depth_gradient_detail.json" 2>/dev/null | tail -1 > "$OUT/bench_c3_depth_gradient.json"
python tools/bench_anchors.py 2>/dev/null | tail -1 > "$OUT/bench_anchors.json"
python tools/parity_report.py c1 c2 c3 c5 dense free_camera precomp lists > "$OUT/parity_report.jsonl" 2>/dev/null
fi
if [ "$PART" != "a" ]; then
kt() {  # name, bench args...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-c4 --no-secondary "$@" > "$OUT/bench_under_rocprof_$name.log" 2>&1 )
  f=$(find "$OUT/kt_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$name.csv"
  rm -rf "$OUT/kt_$name"
}
kt c3 --steps 20 --warmup 5
kt c3_dense --steps 20 --warmup 5 --scale-mul 3
kt c5 --config c5 --steps 10 --warmup 3
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc" > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.json"
python tools/pmc_traffic.py "gpurun_out/$TAG/pmc_summary.json" > "$OUT/pmc_traffic.json"
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc_dense" --scale-mul 3 > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc_dense" > "$OUT/pmc_summary_c3_dense.json"
# keep the merged-back payload small: the raw per-dispatch CSVs stay on the box
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
fi
cut -c1-300 "$OUT/bench_c3_driver_form.json" 2>/dev/null
