OUT=gpurun_out/r6_decode; mkdir -p $OUT; rm -f $OUT/rehearsal2.txt
for L in dec2 dec4; do STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$L.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "codec or gathered" 2>&1 | tail -1 | tee -a $OUT/tests2.txt; done
for rep in 1 2 3; do
  for L in before dec2 dec4; do
    echo "== $L rep $rep" >> $OUT/rehearsal2.txt
    STRSIM_AMD_LIB=$(pwd)/ab_builds/lib$L.so python bench_support/bench_root_rehearsal.py 2>/dev/null | grep -v amdgpu.ids | grep "7 x 14.3 MB\|PIPELINE coded" | tail -2 >> $OUT/rehearsal2.txt
  done
done
cat $OUT/rehearsal2.txt
