#!/bin/bash
# usage: sweep.sh  -- run on the GPU box
B="python bench.py --config cfg5 --rows 6000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
run() { $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],1), 'Mpairs/s', round(d['gcups']), 'GCUPS')"; }
for J in 4 5 6; do
  make -C polars-strsim_amd -B EXTRA=-DSTRSIM_LEV_JOBS=$J >/dev/null 2>&1
  for W in 24 28 32 48; do
    STRSIM_LEV_WAVES_PER_CU=$W run "J=$J W=$W"
  done
done
