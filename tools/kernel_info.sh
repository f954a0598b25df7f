#!/bin/bash
# Register / LDS / code-size figures of the gfx950 kernels of ONE translation unit, from the device assembly
# (no GPU needed).  usage: tools/kernel_info.sh render_bwd [kernel-name-substring] [extra hipcc flags...]
set -e
SRC=$1; PAT=${2:-.}; shift; shift || true
cd "$(dirname "$0")/../bloomscene_amd/csrc"
SLP=-fno-slp-vectorize; [ "$SRC" = preprocess_bwd ] && SLP=
OUT=$(mktemp /tmp/kinfo_XXXX.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None \
  $SLP "$@" --cuda-device-only -S $SRC.hip -o $OUT
python3 - "$OUT" "$PAT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
# instruction counts per function body
bodies = {}
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ins = [l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    bodies[name] = (len(ins), sum(i.startswith('v_') for i in ins), sum(i.startswith('s_') for i in ins),
                    sum(i.startswith('ds_') for i in ins), sum(i.startswith(('global_', 'buffer_', 'flat_')) for i in ins))
for m in re.finditer(r'- \.agpr_count:.*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', txt, re.S):
    name = m.group(1)
    if not re.search(pat, name):
        continue
    blk = m.group(0)
    lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', blk).group(1)
    scr = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1)
    b = bodies.get(name, (0,) * 5)
    print(f"{name[:70]:70s} vgpr {m.group(3):>3s} sgpr {m.group(2):>3s} spill {m.group(4)} lds {lds:>6s} scratch {scr} | static instr {b[0]} (v {b[1]}, s {b[2]}, ds {b[3]}, mem {b[4]})")
PY
rm -f $OUT
