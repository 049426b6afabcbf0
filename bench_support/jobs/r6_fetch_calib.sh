#!/bin/bash
# round 6: what FETCH_SIZE counts for k_lane_wide's access pattern.  bench_support/micro/fetch_calib reads known
# bytes once in four access shapes; the same counters are then taken on cfg3's two kernels.  Each --pmc group is its own run.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6_fetch; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
[ -x bench_support/micro/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 bench_support/micro/fetch_calib.hip -o bench_support/micro/fetch_calib
cd /tmp
rocprofv3 -L > $OUT/avail_all.txt 2>&1
grep -i -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|FETCH_SIZE\|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*\|TCC_REQ_sum\|TCC_MISS_sum\|TCC_HIT_sum" $OUT/avail_all.txt | sort -u > $OUT/avail.txt
$ROOT/bench_support/micro/fetch_calib all > $OUT/calib_plain.txt 2>&1
i=0
for P in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/calib_p$i -- $ROOT/bench_support/micro/fetch_calib all > $OUT/calib_p$i.log 2>&1
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/cfg3_p$i -- python3 $ROOT/bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra-modes > $OUT/cfg3_p$i.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/r6_fetch"
print(open(out + "/calib_plain.txt").read())
for tag in ("calib", "cfg3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/%s_p*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if tag == "cfg3" and not any(x in k for x in ("k_lane_stage", "k_lane_wide")):
                continue
            name = k.split("(")[0][-48:]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s: mean per launch (the LAST launch of each micro kernel is the timed one; both launches read the same bytes)" % tag)
    for k in sorted(acc):
        print("  " + k)
        for c, v in sorted(acc[k].items()):
            print("      %-28s %18.0f   (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
