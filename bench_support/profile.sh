#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repo root):
#   bash bench_support/profile.sh <tag> [bench.py args...]
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, one group per run (TCC FETCH_SIZE and
# WRITE_SIZE cannot share a pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# steady-state recipe: the first ~25 ms after the GPU goes from idle to this load run ~10 % slower
ARGS="--steps ${STEPS:-50} --warmup ${WARMUP:-20} --no-cpu-baseline --no-e2e $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_ta" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_ta.log" 2>&1
cd "$ROOT"
python3 bench_support/summarize_profile.py "$OUT" ${TRAFFIC_KEY:-} ${TRAFFIC_KERNEL:-} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
