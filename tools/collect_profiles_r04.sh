set -u
TAG=${TAG:-r04_v3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench_c3.err" | tail -1 > "$OUT/bench_c3_driver_form.json"
python bench.py --config c2 --steps 100 --warmup 10 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c2.json"
python bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c5.json"
python bench.py --colors precomp --steps 50 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c3_precomp_colors.json"
python bench.py --depth-gradient --steps 50 --warmup 5 --no-cpu-baseline --no-c4 2>/dev/null | tail -1 > "$OUT/bench_c3_depth_gradient.json"
python bench.py --exact-exp --steps 40 --warmup 10 --no-cpu-baseline --no-c4 --no-secondary 2>/dev/null | tail -1 > "$OUT/bench_c3_exact_exp.json"
python bench.py --strict-gradients --steps 40 --warmup 10 --no-cpu-baseline --no-c4 --no-secondary 2>/dev/null | tail -1 > "$OUT/bench_c3_strict_gradients.json"
echo bench done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-secondary > "$OUT/bench_under_rocprof.log" 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_trace_c5" -- python3 "$ROOT/bench.py" --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-c4 > "$OUT/bench_c5_under_rocprof.log" 2>&1 )
echo stats done
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc" > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.json"
python tools/pmc_traffic.py "gpurun_out/$TAG/pmc_summary.json" > "$OUT/pmc_traffic.json"
echo pmc c3 done
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc_c5" --config c5 > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc_c5" > "$OUT/pmc_summary_c5.json"
for d in kernel_trace kernel_trace_c5; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d/kernel_trace/kernel_stats}.csv"; done
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
cut -c1-200 "$OUT/bench_c3_driver_form.json"
