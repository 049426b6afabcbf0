#!/bin/bash
# counters of k_lane_stage for prebuilt library variants: bash bench_support/jobs/pmc_libs.sh "<bench args>" "<counters>" name1 name2 ...
ROOT=$(pwd); export TMPDIR=/tmp
ARGS="$1"; CTRS="$2"; shift 2
for N in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_$N; rm -rf $OUT; mkdir -p $OUT
  export STRSIM_AMD_LIB=$ROOT/ab_builds/lib$N.so
  (cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/bench.py --config cfg2 --rows 20000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e $ARGS > $OUT/log 2>&1)
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$N" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_lane_stage" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], "per 64 rows:", {k: round(sum(v)/len(v)/(20000000/64),1) for k,v in acc.items()})
PY
done
