ROOT=$(pwd); export TMPDIR=/tmp
for V in "-DSTRSIM_LANE_TRIM=0" "-DSTRSIM_LANE_TRIM=1"; do
  make -C polars-strsim_amd -B EXTRA="$V" >/dev/null 2>&1
  OUT=$ROOT/gpurun_out/pmcab_$(echo "$V" | tr -c 'A-Za-z0-9\n' '_'); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $ROOT/bench.py --config cfg2 --rows 20000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/log 2>&1)
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$V" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_lane_pairs" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v)/len(v)/(20000000/64),1) for k,v in acc.items()})
PY
done
make -C polars-strsim_amd -B >/dev/null 2>&1
