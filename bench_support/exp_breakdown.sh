#!/bin/bash
# VALU instructions of k_lane_pairs<levenshtein> with parts of the per-pair work stubbed out (experiment builds only)
ROOT=$(pwd); export TMPDIR=/tmp
for V in "" "-DSTRSIM_EXP_NODP" "-DSTRSIM_EXP_NOPLANES"; do
  make -C polars-strsim_amd -B EXTRA="$V" >/dev/null 2>&1
  OUT=$ROOT/gpurun_out/exp_$(echo "$V" | tr -d '-' | tr -c 'A-Za-z0-9_\n' '_'); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT -- python3 $ROOT/bench.py --config cfg2 --rows 20000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/log 2>&1)
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$V" <<'PY'
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_lane_pairs" in r["Kernel_Name"] and r["Counter_Name"]=="SQ_INSTS_VALU"]
print("variant %-24s VALU per 64 pairs = %.1f" % (sys.argv[2] or "full", sum(v)/len(v)/(20000000/64)))
PY
done
make -C polars-strsim_amd -B >/dev/null 2>&1
