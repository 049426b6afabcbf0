bash bench_support/jobs/pmc_kernel.sh "--config cfg3 --steps 3 --warmup 1" k_lane_wide "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" w1 p1 2>&1 | grep "per launch" | tee gpurun_out/r4_wide_pmc.txt
# counters of k_lane_wide on cfg3 for two prebuilt libraries (names below: the round-4 variants; build_variants.sh)
bash bench_support/jobs/pmc_kernel.sh "--config cfg3 --steps 3 --warmup 1" k_lane_wide "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD" w1 p1 2>&1 | grep "per launch" | tee -a gpurun_out/r4_wide_pmc.txt
