#!/bin/bash
# phase stamps of k_lane_stage for several measures (ab_builds/libstamps.so = EXTRA="-DSTRSIM_LAB -DSTRSIM_STAGE_STAMPS"), then an A/B
#   bash bench_support/jobs/r5_stamps.sh [variants for r5_ab.sh ...]
mkdir -p gpurun_out
{
for m in levenshtein jaro jaccard; do
  STRSIM_AMD_LIB=$(pwd)/ab_builds/libstamps.so python bench_support/stage_stamps.py 20000000 cfg2 $m 2>&1 | head -14
done
} | tee gpurun_out/r5_stage_stamps.txt
[ $# -gt 0 ] && bash bench_support/jobs/r5_ab.sh "$@"
