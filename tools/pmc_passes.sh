#!/bin/bash
# Collect per-kernel hardware counters for bench.py in separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with other trace domains).
# usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-c4 --no-secondary "${BENCH_ARGS[@]}" > "$ROOT/$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum
run grbm GRBM_GUI_ACTIVE
