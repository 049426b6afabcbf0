#!/bin/bash
# rocprofv3 kernel stats of prebuilt library variants: bash bench_support/jobs/trace_libs.sh "<bench args>" name1 name2 ...
ROOT=$(pwd); export TMPDIR=/tmp
ARGS="$1"; shift
for N in "$@"; do
  OUT=$ROOT/gpurun_out/tr_$N; rm -rf $OUT; mkdir -p $OUT
  export STRSIM_AMD_LIB=$ROOT/ab_builds/lib$N.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e $ARGS > $OUT/log 2>&1)
  echo "== $N"; python3 $ROOT/bench_support/summarize_profile.py $OUT 2>/dev/null | grep "strsim::" | head -6 || grep strsim $OUT/*/*kernel_stats.csv | cut -c1-160
done
