#!/bin/bash
# round 6, the headline's one structural experiment: the producer / consumer form of k_lane_stage (csrc/strsim_lane_stage_pc.h, lab
# builds in ab_builds/) against the product kernel, same box, alternating.  bash bench_support/jobs/r6_pc_ab.sh "<lib names>" [measure]
OUT=gpurun_out/${OUTDIR:-r6_pc}; mkdir -p $OUT
LIBS=${1:-"pc1 pc1n pc1n3"}; MEASURE=${2:-levenshtein}
# correctness first: the variant against the oracle (the bench's own 64 M-row comparison + the parity tests that reach this kernel)
for L in $LIBS; do
  echo "== parity, lib$L" >> $OUT/parity.txt
  STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$L.so python bench.py --measure $MEASURE --steps 5 --warmup 2 --no-e2e --no-extra-modes 2>>$OUT/err.txt | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); print("   parity_vs_oracle_on_sample", d.get("parity_vs_oracle_on_sample"), "kernel_ms", d["roofline"]["kernel_ms"])' >> $OUT/parity.txt 2>&1
  STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$L.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$MEASURE and (lane_path_random or row_counts or length_class or reference_vectors or one_launch)" 2>&1 | tail -2 >> $OUT/parity.txt
done
cat $OUT/parity.txt
for rep in 1 2 3; do
  for L in product $LIBS; do
    if [ $L = product ]; then LIB=""; else LIB=$(pwd)/ab_builds/lib$L.so; fi
    STRSIM_AMD_LIB=$LIB python bench.py --measure $MEASURE --steps 30 --warmup 10 --no-cpu-baseline --no-e2e --no-extra-modes 2>/dev/null | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("%-10s %-12s rep '$rep'  %9.1f M/s  %.4f ms/step  kernel %.4f ms  frac %.4f" % ("'$L'", "'$MEASURE'", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"]))' | tee -a $OUT/ab_$MEASURE.txt
  done
done
