#!/bin/bash
# Round-5 sanity run after a kernel change (one gpurun call): the GPU suite on the tree's library, then the A/B of the variants given.
OUT=gpurun_out/r5_check; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
[ $# -gt 0 ] && bash bench_support/jobs/r5_ab.sh "$@"
