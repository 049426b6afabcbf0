#!/bin/bash
# GPU suite of the in-tree build, then the same-box A/B of prebuilt libraries (cfg3 + the 33..128-byte frame)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4_full_pytest.txt
bash bench_support/jobs/r4_ab2.sh "$@"
