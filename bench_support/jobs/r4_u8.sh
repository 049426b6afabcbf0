#!/bin/bash
mkdir -p gpurun_out
{
for N in u0 u1; do echo "== non-ascii $N"; STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$N.so python bench_support/bench_non_ascii.py 2>&1 | grep -v amdgpu.ids | tail -8; done
for N in u0 u1; do echo "== non-ascii $N (again)"; STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$N.so python bench_support/bench_non_ascii.py 2>&1 | grep -v amdgpu.ids | tail -8; done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "utf8 or script or non_ascii or unicode" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r4_u8.txt
