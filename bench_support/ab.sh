#!/bin/bash
# A/B on one box: bash bench_support/ab.sh "<EXTRA flags A>" "<EXTRA flags B>" [bench args]   (cfg2 lane-kernel ms)
A="$1"; B="$2"; shift 2
for round in 1 2; do
for V in "$A" "$B"; do
  make -C polars-strsim_amd -B EXTRA="$V" >/dev/null 2>&1
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-40s value %8.1f  lane_ms %.4f  wave_ms %.4f' % ('[$V]', d['value'], r['kernel_ms'], r['wave_kernel_ms']))"
done; done
make -C polars-strsim_amd -B >/dev/null 2>&1
