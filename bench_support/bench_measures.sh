#!/bin/bash
# the five measures on the cfg2 frame (100 M rows, U{1..32}, a-z), one line each: bash bench_support/bench_measures.sh [ENV=..]
for m in levenshtein jaro jaro_winkler jaccard sorensen_dice; do
  env "$@" python bench.py --measure $m --steps 20 --warmup 10 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-14s %8.1f M pairs/s  ms/step %.3f lane_ms %.3f frac %.3f' % ('$m', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))"
done
