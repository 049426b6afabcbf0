#!/bin/bash
# same-box A/B of prebuilt library variants (bench_support/build_variants.sh):
#   bash bench_support/jobs/ab_libs.sh "<bench args>" name1[:WG_PER_CU] name2 ...
ARGS="$1"; shift
for round in 1 2; do
for SPEC in "$@"; do
  N=${SPEC%%:*}
  if [[ "$SPEC" == *:* ]]; then export STRSIM_STAGE_WG_PER_CU=${SPEC##*:}; else unset STRSIM_STAGE_WG_PER_CU; fi
  STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$N.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e $ARGS 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-14s value %8.1f  lane_ms %.4f  wave_ms %.4f  ms/step %.4f' % ('$SPEC', d['value'], r['kernel_ms'], r['wave_kernel_ms'], d['ms_per_step']))"
done; done
