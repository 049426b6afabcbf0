#!/bin/bash
# same-box scan of the plugin pipeline's knobs: bash bench_support/e2e_scan.sh "<ENV=.. ENV=..>" "<...>" ...
for rep in 1 2; do
for e in "$@"; do
  env $e POLARS_STRSIM_TRACE=1 python bench_support/bench_plugin_e2e.py 10000000 vu 2>&1 | grep -v amdgpu.ids | grep "rows=10000000\|levenshtein:" | sort -t= -k3 -n | awk -v tag="[$e]" '/rows=10000000/{ if (!best) best=$0 } /levenshtein:/{lev=$0} END{print tag; print "   " best; print "   " lev}'
done; done
