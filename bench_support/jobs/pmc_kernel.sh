#!/bin/bash
# counters of one kernel for prebuilt library variants (ONE group of counters per call that the hardware can collect in one pass:
# "FETCH_SIZE WRITE_SIZE TA_TA_BUSY_sum" together hung a box for the whole time limit in round 4 -- bench_support/profile.sh
# takes them in separate passes):
#   bash bench_support/jobs/pmc_kernel.sh "<bench args>" "<kernel substring>" "<counters>" name1 name2 ...
ROOT=$(pwd); export TMPDIR=/tmp
ARGS="$1"; KERN="$2"; CTRS="$3"; shift 3
for N in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_$N; rm -rf $OUT; mkdir -p $OUT
  export STRSIM_AMD_LIB=$ROOT/ab_builds/lib$N.so
  (cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e $ARGS > $OUT/log 2>&1)
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$N" "$KERN" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[3] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], "per launch:", {k: "%.4g" % (sum(v)/len(v)) for k,v in acc.items()}, "launches", {k: len(v) for k,v in acc.items()})
PY
done
