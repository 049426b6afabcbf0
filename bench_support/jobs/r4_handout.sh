#!/bin/bash
# same-box A/B of the round hand-out of k_lane_stage (0 static, 1 rotated, 2 counter): cfg2, cfg3, cfg2 Jaro-Winkler
mkdir -p gpurun_out
{
echo "== cfg2 levenshtein"; bash bench_support/jobs/ab_libs.sh "--config cfg2" h0 h2 h3
echo "== cfg3"; bash bench_support/jobs/ab_libs.sh "--config cfg3" h0 h2 h3
echo "== cfg2 jaro_winkler"; bash bench_support/jobs/ab_libs.sh "--config cfg2 --measure jaro_winkler" h0 h2 h3
} 2>&1 | tee gpurun_out/r4_handout.txt
