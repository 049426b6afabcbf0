#!/bin/bash
# round 6: the root's decode of a gathered column with 16-byte stores (two rows per thread) against the 8-byte form (ab_builds/libbefore.so)
OUT=gpurun_out/r6_decode; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "codec or gathered" 2>&1 | tail -2 | tee $OUT/tests.txt
python -m pytest tests/test_gpu_multirank_smoke.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT/tests.txt
for rep in 1 2 3; do
  for L in before product; do
    if [ $L = product ]; then LIB=""; else LIB=$(pwd)/ab_builds/lib$L.so; fi
    echo "== $L rep $rep" >> $OUT/rehearsal.txt
    STRSIM_AMD_LIB=$LIB python bench_support/bench_root_rehearsal.py 2>/dev/null | grep -v amdgpu.ids | grep "decode_gathered alone\|PIPELINE coded" >> $OUT/rehearsal.txt
  done
done
cat $OUT/rehearsal.txt
STRSIM_AMD_LIB= python bench_support/bench_root_rehearsal.py 2>/dev/null | grep -v amdgpu.ids > $OUT/root_rehearsal_final.txt
