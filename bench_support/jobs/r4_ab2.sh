#!/bin/bash
# same-box A/B of two prebuilt libraries on cfg3 (+ the 33..128-byte frame): bash bench_support/jobs/r4_ab2.sh name0 name1 ...
mkdir -p gpurun_out
{
echo "== cfg3"; bash bench_support/jobs/ab_libs.sh "--config cfg3" "$@"
for N in "$@"; do echo "== mid ascii $N"; STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$N.so python bench_support/bench_mid_ascii.py 2>&1 | tail -6; done
} 2>&1 | tee gpurun_out/r4_ab2.txt
