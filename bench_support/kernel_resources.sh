#!/bin/bash
# Registers / LDS / scratch of the kernels in csrc/strsim_kernels.hip (device-only compile, then the code object's metadata).
#   bash bench_support/kernel_resources.sh [name-filter-regex] [EXTRA flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=${TMPDIR:-/tmp}/strsim_co; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$ROOT/include -I$ROOT/polars-strsim_amd/csrc $2 \
  --cuda-device-only -c -x hip $ROOT/polars-strsim_amd/csrc/strsim_kernels.hip -o $OUT/k.co 2>/dev/null || { echo "compile failed"; exit 1; }
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$OUT/k.co --output=$OUT/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT/k.elf > $OUT/notes.txt
python3 - "$OUT/notes.txt" "${1:-.}" <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
for e in re.split(r'\n\s+- \.agpr_count:', t)[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', e) or [None, '?'])[1]
    n = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    if re.search(sys.argv[2], n):
        print('%-64s vgpr %4s sgpr %4s lds %6s scratch %5s spill %s' % (n[:64], g('vgpr_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count')))
PY
