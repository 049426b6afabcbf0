#!/bin/bash
# round 6, first check of the staging pool: the new tests, the plugin ABI suites, the lone 10 M-row call against the round-start library
mkdir -p gpurun_out/r6b
python -m pytest tests/test_plugin_staging_gpu.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r6b/staging.txt
python -m pytest tests/test_plugin_abi_gpu.py tests/test_plugin_configs_gpu.py -m gpu -q 2>&1 | tail -8 > gpurun_out/r6b/plugin.txt
python -m pytest tests/test_knobs_gpu.py -m gpu -q -k "STAGING or VIEWS" 2>&1 | tail -5 >> gpurun_out/r6b/plugin.txt
for rep in 1 2 3; do
  for lib in ab_builds/libr6base.so ""; do
    echo "== lib=${lib:-product} rep $rep" >> gpurun_out/r6b/e2e_ab.txt
    STRSIM_AMD_LIB=$lib bash bench_support/jobs/plugin_e2e.sh 2>/dev/null | grep "^==" >> gpurun_out/r6b/e2e_ab.txt
  done
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r6b/bench_cfg2.json 2> gpurun_out/r6b/bench_cfg2.err
