#!/bin/bash
# builds library variants for same-box A/B runs: bash bench_support/build_variants.sh name1 "flags1" name2 "flags2" ...
# -> ab_builds/lib<name>.so (selected at run time with STRSIM_AMD_LIB)
mkdir -p ab_builds
while [ $# -ge 2 ]; do
  N="$1"; F="$2"; shift 2
  make -C polars-strsim_amd -B EXTRA="$F" OUT=$(pwd)/ab_builds/lib$N.so 2>&1 | grep -i "error" 
  ls -la ab_builds/lib$N.so
done
