#!/bin/bash
# Build the library of a git revision (default HEAD) next to the working tree's, for tools/ab_bench.sh:
#   tools/build_base.sh [rev] [name]  ->  bloomscene_amd/libbsr_<name>.so   (name defaults to "base")
set -e
REV=${1:-HEAD}; NAME=${2:-base}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$ROOT/build/$NAME" && mkdir -p "$ROOT/build/$NAME"
git -C "$ROOT" archive "$REV" bloomscene_amd/csrc include | tar -x -C "$ROOT/build/$NAME"
make -C "$ROOT/build/$NAME/bloomscene_amd/csrc" -j8 > /dev/null
cp "$ROOT/build/$NAME/bloomscene_amd/libbloomscene_rast.so" "$ROOT/bloomscene_amd/libbsr_$NAME.so"
rm -rf "$ROOT/build/$NAME"
echo "bloomscene_amd/libbsr_$NAME.so = $REV"
