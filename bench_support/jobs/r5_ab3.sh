#!/bin/bash
# cfg3 + the 33..128-byte frame only: bash bench_support/jobs/r5_ab3.sh name0 name1 ...
mkdir -p gpurun_out
TAG=$(echo "$@" | tr ' :' '__')
{
echo "== cfg3 (jaro_winkler, Zipf 4..128)"; bash bench_support/jobs/ab_libs.sh "--config cfg3" "$@"
echo "== cfg3 lengths, jaro"; bash bench_support/jobs/ab_libs.sh "--config cfg3 --measure jaro" "$@"
for N in "$@"; do echo "== mid ascii ${N%%:*}"; STRSIM_AMD_LIB=$(pwd)/ab_builds/lib${N%%:*}.so python bench_support/bench_mid_ascii.py 2>&1 | tail -5; done
} 2>&1 | tee gpurun_out/r5_ab3_$TAG.txt
