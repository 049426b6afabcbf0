for round in 1 2; do
for cfg in "6 128" "7 128" "8 128"; do
  set -- $cfg
  make -C polars-strsim_amd -B EXTRA="-DSTRSIM_LANE_WAVES_PER_EU=$1" >/dev/null 2>&1
  for m in levenshtein jaro jaccard; do
  STRSIM_LANE_WG_PER_CU=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --measure $m 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('waves_per_eu=$1 wg_per_cu=$2 %-12s value %8.1f  lane_ms %.4f' % ('$m', d['value'], r['kernel_ms']))"
  done
done; done
make -C polars-strsim_amd -B >/dev/null 2>&1
