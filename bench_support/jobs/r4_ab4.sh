#!/bin/bash
# same-box A/B of prebuilt libraries: cfg2 Jaro / Jaro-Winkler / all five, cfg3
mkdir -p gpurun_out
{
echo "== cfg2 jaro"; bash bench_support/jobs/ab_libs.sh "--config cfg2 --measure jaro" "$@"
echo "== cfg2 jaro_winkler"; bash bench_support/jobs/ab_libs.sh "--config cfg2 --measure jaro_winkler" "$@"
echo "== all, 100 M rows"; bash bench_support/jobs/ab_libs.sh "--measure all --rows 100000000" "$@"
echo "== cfg3"; bash bench_support/jobs/ab_libs.sh "--config cfg3" "$@"
} 2>&1 | tee gpurun_out/r4_ab4.txt
