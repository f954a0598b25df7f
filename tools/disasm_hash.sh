#!/bin/bash
# usage: disasm_hash.sh file.hip [extra flags] -> sha256 of the gfx950 device disassembly (text only)
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize "$@" --cuda-device-only -S -o - "$src" 2>/dev/null | grep -v '^\s*[;.]' | grep -v '^\s*$' | sha256sum
