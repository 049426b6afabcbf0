#!/bin/bash
OUT=gpurun_out/r4_f1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -k "column_from_views or pipelined_caller or one_launch or retire" -x -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_plugin_abi_gpu.py -x -q 2>&1 | tail -5
python bench_support/bench_views.py 10000000 2>$OUT/err.txt | tee $OUT/f1_views.txt
tail -3 $OUT/err.txt
