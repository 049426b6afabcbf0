#!/bin/bash
# Round-5 same-box A/B of prebuilt library variants (ab_builds/lib<name>.so) over the Jaro family's workloads:
#   bash bench_support/jobs/r5_ab.sh name0 name1 ...     -> gpurun_out/r5_ab_<names>.txt
mkdir -p gpurun_out
TAG=$(echo "$@" | tr ' :' '__')
{
echo "== cfg3 (jaro_winkler, Zipf 4..128)"; bash bench_support/jobs/ab_libs.sh "--config cfg3" "$@"
echo "== cfg2 lengths, jaro"; bash bench_support/jobs/ab_libs.sh "--measure jaro" "$@"
echo "== cfg2 lengths, jaro_winkler"; bash bench_support/jobs/ab_libs.sh "--measure jaro_winkler" "$@"
echo "== five outputs, 100 M rows"; bash bench_support/jobs/ab_libs.sh "--measure all --rows 100000000" "$@"
for N in "$@"; do echo "== mid ascii ${N%%:*}"; STRSIM_AMD_LIB=$(pwd)/ab_builds/lib${N%%:*}.so python bench_support/bench_mid_ascii.py 2>&1 | tail -5; done
} 2>&1 | tee gpurun_out/r5_ab_$TAG.txt
