#!/bin/bash
# cfg5 (k_wave_pairs<levenshtein>, 64 pattern rows per lane) at the occupancy a 128-rows-per-lane form would have: its split match
# tables double (12 entries x 4 mask words x 256 B = 12 KB per wave instead of 6: 7.6 -> 13.7 KB of LDS per wave, 20 -> 11 waves per CU).
# The form's whole gain is bounded by its instruction saving (49 instructions per 128 cells against 27 per 64: -9 %);
# what the same kernel loses at 11 waves per CU is the other side of that trade.  STRSIM_LEV_WAVES_PER_CU caps the persistent grid.
mkdir -p gpurun_out
{
for w in 20 16 13 11; do
  STRSIM_LEV_WAVES_PER_CU=$w python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('waves per CU %2s: %7.1f M pairs/s  %6.2f ms/step  k_wave_pairs %.2f ms  (%.1f TCUPS)' % ('$w', d['value'], d['ms_per_step'], r['wave_kernel_ms'], d['gcups'] / 1e3))"
done
} 2>&1 | tee gpurun_out/r5_cfg5_occupancy.txt
