#!/bin/bash
mkdir -p gpurun_out
{
bash bench_support/jobs/ab_libs.sh "--config cfg3" q2 x1
bash bench_support/jobs/pmc_kernel.sh "--config cfg3 --steps 3 --warmup 1" k_lane_wide "FETCH_SIZE WRITE_SIZE TA_TA_BUSY_sum" q2 x1 2>&1 | grep "per launch"
} 2>&1 | tee gpurun_out/r4_x1.txt
