set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_na; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench_support/bench_non_ascii.py 4000000 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench_support/bench_non_ascii.py 4000000 > $OUT/pmc.log 2>&1
cd $ROOT; python3 bench_support/summarize_profile.py $OUT | grep -v "at::\|rocprim\|rocclr" | head -60; tail -7 $OUT/trace.log
