#!/bin/bash
# PMC passes over the one-pair-per-lane kernel of one build/arm (GPU box): bash bench_support/pmc_lane.sh <tag> "<ENV=..>" [bench args]
# prints the mean per launch of each counter for the one-pair-per-lane kernels (k_lane_pairs, k_lane_stage)
TAG=$1; ARM="$2"; shift 2
ROOT=$(pwd); export TMPDIR=/tmp
ROWS=${ROWS:-20000000}
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"
 "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "SQ_IFETCH SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_SMEM"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1)); OUT=$ROOT/gpurun_out/pmc_${TAG}/p$i; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && env $ARM rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -- python3 $ROOT/bench.py --config cfg2 --rows $ROWS --steps 3 --warmup 2 --no-cpu-baseline --no-e2e "$@" > $OUT/log 2>&1)
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$ROWS" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if any(k in r["Kernel_Name"] for k in ("k_lane_pairs", "k_lane_stage")): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("  (no counters: %r)" % (e,))
rows=float(sys.argv[2])
for k,v in acc.items():
    m=sum(v)/len(v)
    print("  %-36s %16.0f   per 64 pairs %10.2f" % (k, m, m/(rows/64)))
PY
done 2>&1 | tee $ROOT/gpurun_out/pmc_${TAG}/summary.txt
