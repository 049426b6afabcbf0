#!/bin/bash
# round 6: k_lane_wide with the next two-word round's windows fetched into registers under the current round's
# cores (lab builds wpf1: pattern + text, wpf2: pattern only; -DSTRSIM_WIDE_PREFETCH=1|2), same box, alternating.
OUT=gpurun_out/r6_wpf; mkdir -p $OUT
LIBS=${1:-"wpf1 wpf2"}
for L in product $LIBS; do
  if [ $L = product ]; then LIB=""; else LIB=$(pwd)/ab_builds/lib$L.so; fi
  echo "== parity, $L" >> $OUT/parity.txt
  STRSIM_AMD_LIB=$LIB timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or length_class or mixed or beyond or random" 2>&1 | tail -2 >> $OUT/parity.txt
done
STRSIM_AMD_LIB=$(pwd)/ab_builds/libwpf1.so timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg3" 2>&1 | tail -2 >> $OUT/parity.txt
cat $OUT/parity.txt
for rep in 1 2 3; do
  for L in product w1t32 $LIBS; do
    if [ $L = product ]; then LIB=""; else LIB=$(pwd)/ab_builds/lib$L.so; fi
    STRSIM_AMD_LIB=$LIB python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-extra-modes 2>/dev/null | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("%-10s cfg3 rep '$rep'  %9.1f M/s  %.4f ms/step  stage %.4f + wide %.4f ms  frac %.4f" % ("'$L'", d["value"], d["ms_per_step"], r["kernel_ms"], r["wave_kernel_ms"], r["frac"]))' | tee -a $OUT/ab_cfg3.txt
  done
done
for L in product $LIBS; do
  if [ $L = product ]; then LIB=""; else LIB=$(pwd)/ab_builds/lib$L.so; fi
  echo "== 33..128-byte frame, $L" >> $OUT/mid_ascii.txt
  STRSIM_AMD_LIB=$LIB python bench_support/bench_mid_ascii.py 2>/dev/null | tail -6 >> $OUT/mid_ascii.txt
done
cat $OUT/mid_ascii.txt
