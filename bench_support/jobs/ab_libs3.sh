#!/bin/bash
# as ab_libs.sh, three alternations and --steps 20: bash bench_support/jobs/ab_libs3.sh "<bench args>" name1 name2 ...
ARGS="$1"; shift
for round in 1 2 3; do
for SPEC in "$@"; do
  N=${SPEC%%:*}
  STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$N.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e $ARGS 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-14s value %8.1f  lane_ms %.4f  wave_ms %.4f  ms/step %.4f' % ('$SPEC', d['value'], r['kernel_ms'], r['wave_kernel_ms'], d['ms_per_step']))"
done; done
