#!/bin/bash
# round 6: aggregate rate of concurrent small calls (is there anything for a coalescer to win?) + the staging test after the estimate change
OUT=gpurun_out/r6_small; mkdir -p $OUT
g++ -O2 -std=c++17 -pthread -Iinclude bench_support/micro/small_call_threads.cpp -o bench_support/micro/small_call_threads polars-strsim_amd/polars_strsim/libpolars_strsim_amd.so -Wl,-rpath,$PWD/polars-strsim_amd/polars_strsim
bench_support/micro/small_call_threads 3000 2>/dev/null | tee $OUT/small_call_threads.txt
python -m pytest tests/test_plugin_staging_gpu.py -m gpu -q 2>&1 | tail -3 | tee $OUT/staging_test.txt
POLARS_STRSIM_STAGING_BUDGET_MB=1024 python tests/helpers/staging_child.py 32 4000000 2>/dev/null | tee $OUT/staging_1gib.json
python tests/helpers/staging_child.py 32 4000000 2>/dev/null | tee $OUT/staging_default.json
