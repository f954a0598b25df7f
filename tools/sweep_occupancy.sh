#!/bin/bash
# BASELINE config C5, "LDS-tile occupancy + rocprof HBM-GB/s sweep": the two tile renderers at 5 M Gaussians with
#   * (until round 5 also the network walk at 64 / 128 / 256 staged entries: profiles/r02-r04), and
#   * the workgroups a CU can hold capped by unused dynamic LDS (BSR_SWEEP_LDS_PAD_FWD / _BWD, read only by the sweep
#     build libbsr_rast_sweep.so: make -C bloomscene_amd/csrc sweep; tile_common.h),
# each point: bench.py --config c5 (stage times from hipEvents) + rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
# in separate passes, SQ counters in a third) -> tools/sweep_occupancy.py prints one JSON line per point.
# usage (GPU box, repo root): bash tools/sweep_occupancy.sh <outdir>
set -u
OUT=${1:-gpurun_out/sweep_occ}
CONFIG=${2:-c5}      # c5 (BASELINE's sweep) or c3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
# name lib pad_fwd pad_bwd
# (round 5: every non-strict frame takes the transposed walk, 31.0 KB LDS -> the same pads at C3 and C5: 5000 / 16000 / 36000
#  bytes leave 4 / 3 / 2 workgroups per CU; forward 20.3 KB: 6500 / 20100 / 61000 leave 6 / 4 / 2)
POINTS=(
 "native            libbsr_rast_sweep.so  0     0"
 "bwd_wg4           libbsr_rast_sweep.so  0     5000"
 "bwd_wg3           libbsr_rast_sweep.so  0     16000"
 "bwd_wg2           libbsr_rast_sweep.so  0     36000"
 "fwd_wg6           libbsr_rast_sweep.so  6500  0"
 "fwd_wg4           libbsr_rast_sweep.so  20100 0"
 "fwd_wg2           libbsr_rast_sweep.so  61000 0"
)
for p in "${POINTS[@]}"; do
  set -- $p
  name=$1; lib=$2; export BSR_SWEEP_LDS_PAD_FWD=$3; export BSR_SWEEP_LDS_PAD_BWD=$4
  [ -f "$ROOT/bloomscene_amd/$lib" ] || { echo "skip $name: $lib missing"; continue; }
  LIBARG="--lib $ROOT/bloomscene_amd/$lib"
  d=$ROOT/$OUT/$name; mkdir -p "$d"
  ( cd "$ROOT" && python bench.py $LIBARG --config $CONFIG --steps 10 --warmup 3 --no-cpu-baseline --no-c4 --no-secondary 2>/dev/null | tail -1 > "$d/bench.json" )
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "grbm GRBM_GUI_ACTIVE"; do
    set -- $pass; pn=$1; shift
    ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$d/$pn" -- python3 "$ROOT/bench.py" $LIBARG --config $CONFIG --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-c4 --no-secondary > "$d/$pn.log" 2>&1 )
  done
  python "$ROOT/tools/pmc_summary.py" "$d" > "$d/pmc_summary.json"
  find "$d" -name "*.csv" -delete
  echo "$name done" >&2
done
python "$ROOT/tools/sweep_occupancy.py" "$ROOT/$OUT"
