#!/bin/bash
# A/B on one box with ONE build: bash bench_support/ab_env.sh "<ENV=..> [ENV=..]" "<ENV=..>" ... -- [bench args]
# each quoted argument is a set of environment assignments for one arm; two rounds, arms interleaved
ARMS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for round in 1 2; do
for V in "${ARMS[@]}"; do
  env $V python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-44s value %8.1f  ms/step %.4f lane_ms %.4f  wave_ms %.4f frac %.3f wave_rows %s' % ('[$V]', d['value'], d['ms_per_step'], r['kernel_ms'], r['wave_kernel_ms'], r['frac'], d['config']['rows_on_wave_kernel']))"
done; done
