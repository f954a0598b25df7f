#!/bin/bash
# SQ counters of the tile walks for ONE library build (A/B runs): tools/pmc_ab.sh <outdir> <lib.so> [bench args...]
# (rocprofv3 --pmc passes with --kernel-trace only; the program itself after "--")
set -u
OUT=$1; LIB=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/bench.py" --lib "$ROOT/$LIB" --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-c4 --no-secondary "${BENCH_ARGS[@]}" > "$ROOT/$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.json"
