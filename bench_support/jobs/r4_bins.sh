#!/bin/bash
# binned path: its tests first (tables and bit fills), then kernel traces of cfg3 with tables / with bit fills / unbinned
OUT=gpurun_out/r4_bins; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bins.py -x -q > $OUT/pytest_bins.txt 2>&1; tail -5 $OUT/pytest_bins.txt
STRSIM_BINS_LUT=0 timeout 900 python -m pytest tests/test_gpu_bins.py -x -q > $OUT/pytest_bins_nolut.txt 2>&1; tail -3 $OUT/pytest_bins_nolut.txt
ROOT=$(pwd); export TMPDIR=/tmp
for v in lut nolut; do
  if [ $v = nolut ]; then export STRSIM_BINS_LUT=0; else export STRSIM_BINS_LUT=1; fi
  ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_$v" -- python3 "$ROOT/bench.py" --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > "$ROOT/$OUT/trace_$v.log" 2>&1 )
  echo "== $v"
  python3 - $OUT/trace_$v <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'strsim' in r['Name'] and float(r['AverageNs'])>20000:
        print('%-44s calls %3s avg %10.1f us' % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
unset STRSIM_BINS_LUT
for envs in "A=1" "STRSIM_NO_BINS=1" "STRSIM_BINS_LUT=0"; do
  env $envs python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>>$OUT/bench.err | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$envs  %9.1f M/s  %8.4f ms/step  kernel %.4f + %.4f ms  ops/step %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['wave_kernel_ms'], d['config']['enqueued_kernels_and_copies_per_step']))"
done
