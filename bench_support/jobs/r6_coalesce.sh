#!/bin/bash
# round 6: concurrent small calls combined into one launch -- tests, then the plugin-ABI rate with and without (C++ threads, no Python)
OUT=gpurun_out/r6_coalesce; mkdir -p $OUT; rm -f $OUT/plugin_small_threads.txt
python -m pytest tests/test_plugin_coalesce_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee $OUT/tests.txt
g++ -O2 -std=c++17 -pthread -Iinclude bench_support/micro/plugin_small_threads.cpp -o bench_support/micro/plugin_small_threads polars-strsim_amd/polars_strsim/libpolars_strsim_amd.so -Wl,-rpath,$PWD/polars-strsim_amd/polars_strsim
for mode in ${MODES:-"POLARS_STRSIM_COALESCE=0" "POLARS_STRSIM_COALESCE=1" "POLARS_STRSIM_COALESCE_MIN_INFLIGHT=12"}; do
  echo "== $mode" | tee -a $OUT/plugin_small_threads.txt
  env $mode bench_support/micro/plugin_small_threads 2000 2>/dev/null | tee -a $OUT/plugin_small_threads.txt
done
