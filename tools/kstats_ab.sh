#!/bin/bash
# Kernel-trace statistics of several builds of the library on ONE GPU box:
#   bash tools/kstats_ab.sh <tag> "<bench args>" libA.so libB.so ...      ("product" = the in-tree library)
# -> gpurun_out/<tag>/<lib>.kernel_stats.csv + a one-line-per-kernel summary on stdout (avg us per launch).
set -u
TAG=$1; ARGS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for L in "$@"; do
  name=$(basename "$L" .so)
  libarg=""; [ "$L" != "product" ] && libarg="--lib $ROOT/$L"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -- python3 "$ROOT/bench.py" $libarg --no-cpu-baseline --no-c4 --no-secondary $ARGS > "$OUT/$name.bench.log" 2>&1 )
  f=$(find "$OUT/kt_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/$name.kernel_stats.csv"
  rm -rf "$OUT/kt_$name"
  echo "== $name: $(tail -1 "$OUT/$name.bench.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" 2>/dev/null)"
  python3 - "$OUT/$name.kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    if int(r["Calls"]) < 10:
        continue
    n = r["Name"].split("(")[0].replace("void ", "")[:44]
    print("   %-44s calls %5s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
