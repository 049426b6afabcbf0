#!/bin/bash
# parity of the wide / mid-length classes, then the same-box A/B of two libraries (cfg3 + the 33..128-byte frame)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_hypothesis.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4_wide_pytest.txt
bash bench_support/jobs/r4_ab2.sh "$@"
