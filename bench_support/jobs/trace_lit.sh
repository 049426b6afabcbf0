#!/bin/bash
# kernel durations of the literal path for two library builds (rocprofv3 --kernel-trace --stats)
ROOT=$(pwd); export TMPDIR=/tmp
for N in "$@"; do
  OUT=$ROOT/gpurun_out/trace_lit_$N; rm -rf $OUT; mkdir -p $OUT
  export STRSIM_AMD_LIB=$ROOT/ab_builds/lib$N.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench_support/bench_literal.py > $OUT/log 2>&1)
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $N"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "strsim" in r["Name"]: print("%-60s calls %5s avg %9.1f us min %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
