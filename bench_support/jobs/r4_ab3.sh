#!/bin/bash
# same-box A/B of prebuilt libraries on cfg2 (Levenshtein, Jaro-Winkler, Jaccard), cfg3 and cfg4's pass
mkdir -p gpurun_out
{
echo "== cfg2 levenshtein"; bash bench_support/jobs/ab_libs.sh "--config cfg2" "$@"
echo "== cfg2 jaro_winkler"; bash bench_support/jobs/ab_libs.sh "--config cfg2 --measure jaro_winkler" "$@"
echo "== cfg2 jaccard"; bash bench_support/jobs/ab_libs.sh "--config cfg2 --measure jaccard" "$@"
echo "== cfg3"; bash bench_support/jobs/ab_libs.sh "--config cfg3" "$@"
} 2>&1 | tee gpurun_out/r4_ab3.txt
