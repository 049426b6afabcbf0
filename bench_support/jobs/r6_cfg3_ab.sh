#!/bin/bash
# round 6: cfg3 with the rows whose SHORTER string exceeds T bytes left to k_lane_wide's one-word class
# (lab builds w1t<T> in ab_builds/: -DSTRSIM_WIDE_ONE_WORD=1 -DSTRSIM_STAGE_LONG_TEXT_MAX=<T>), same box, alternating.
OUT=gpurun_out/r6_cfg3; mkdir -p $OUT
LIBS=${1:-"w1t32 w1t24 w1t20 w1t16"}
for L in $LIBS; do
  echo "== parity, lib$L" >> $OUT/parity.txt
  STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$L.so timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or length_class or mixed or beyond or random" 2>&1 | tail -2 >> $OUT/parity.txt
done
STRSIM_AMD_LIB=$(pwd)/ab_builds/libw1t20.so timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg3" 2>&1 | tail -2 >> $OUT/parity.txt
cat $OUT/parity.txt
for rep in 1 2 3; do
  for L in product $LIBS; do
    if [ $L = product ]; then LIB=""; else LIB=$(pwd)/ab_builds/lib$L.so; fi
    STRSIM_AMD_LIB=$LIB python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-extra-modes 2>/dev/null | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("%-10s cfg3 rep '$rep'  %9.1f M/s  %.4f ms/step  stage %.4f + wide %.4f ms  frac %.4f  rows_on_wave_kernel %s" % ("'$L'", d["value"], d["ms_per_step"], r["kernel_ms"], r["wave_kernel_ms"], r["frac"], d["config"]["rows_on_wave_kernel"]))' | tee -a $OUT/ab_cfg3.txt
  done
done
