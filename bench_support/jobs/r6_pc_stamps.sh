#!/bin/bash
# round 6: where the producer / consumer form of k_lane_stage spends a wave's life (lab builds with phase stamps), + the FETCH_SIZE calibration
OUT=gpurun_out/r6_pc; mkdir -p $OUT
STRSIM_AMD_LIB=$(pwd)/ab_builds/libpc1s.so python bench_support/stage_pc_stamps.py 100000000 levenshtein 2 2>/dev/null > $OUT/stamps_pc1_lut_2wg.txt
STRSIM_AMD_LIB=$(pwd)/ab_builds/libpc1n3s.so python bench_support/stage_pc_stamps.py 100000000 levenshtein 3 2>/dev/null > $OUT/stamps_pc1n3_fills_3wg.txt
cat $OUT/stamps_*.txt
bash bench_support/jobs/r6_fetch_calib.sh > gpurun_out/r6_fetch_summary.txt 2>&1
tail -80 gpurun_out/r6_fetch_summary.txt
