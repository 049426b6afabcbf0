#!/bin/bash
# Round-6 evidence run (one gpurun call): counter profiles of the FINAL build first (they write profiles/traffic.json, which the bench
# lines taken afterwards replay -- the line says so in roofline.traffic_source), then the bench lines of every config and measure, the
# side benches, the one-GPU stand-ins for N > 1, the GPU suite.
OUT=gpurun_out/r6_final; rm -rf $OUT; mkdir -p $OUT
STEPS=150 WARMUP=20 TRAFFIC_KEY=cfg2:levenshtein:100000000 TRAFFIC_KERNEL=k_lane_stage bash bench_support/profile.sh r6_cfg2 --no-extra-modes > $OUT/prof_cfg2.txt 2>&1
STEPS=8 WARMUP=3 TRAFFIC_KEY=cfg3:jaro_winkler:100000000 TRAFFIC_KERNEL=k_lane_stage+k_lane_wide bash bench_support/profile.sh r6_cfg3 --config cfg3 --no-extra-modes > $OUT/prof_cfg3.txt 2>&1
STEPS=3 WARMUP=1 TRAFFIC_KEY=cfg5:levenshtein:10000000 TRAFFIC_KERNEL=k_lane_stage+k_wave_pairs bash bench_support/profile.sh r6_cfg5 --config cfg5 --no-extra-modes > $OUT/prof_cfg5.txt 2>&1
cp profiles/traffic.json $OUT/traffic.json
python bench.py > $OUT/bench_default.jsonl 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.jsonl | cut -c1-600
tail -1 $OUT/bench_default.jsonl > $OUT/bench_lines.jsonl
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> $OUT/bench_lines.jsonl   # (the driver's flags)
for args in "--config cfg1 --steps 200" "--config cfg3 --steps 10 --warmup 3" "--config cfg5 --steps 3 --warmup 1" "--measure all --rows 100000000 --steps 10 --warmup 3" "--measure jaro" "--measure jaro_winkler" "--measure jaccard" "--measure sorensen_dice"; do
  python bench.py $args --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 >> $OUT/bench_lines.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r6_final/bench_lines.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print("%-64s %9.1f M/s %8.4f ms/step kernel %.4f + %.4f ms frac %.4f traffic %s | default mode %s cold %s idle %s ops/step %s %s" % (
        d["metric"][:64], d["value"], d["ms_per_step"], r["kernel_ms"], r["wave_kernel_ms"], r["frac"], r["traffic"], d.get("value_default_mode"),
        d.get("cold_first_call_ms"), d.get("idle_gpu_call_ms"), d["config"]["enqueued_kernels_and_copies_per_step"], d.get("gcups", "")))
PY
python bench_support/jobs/small_frames.py > $OUT/small_frames.txt 2>/dev/null
python bench_support/bench_literal.py > $OUT/literal.txt 2>/dev/null
python bench_support/bench_mid_ascii.py 2>/dev/null | tail -5 > $OUT/mid_ascii.txt
bash bench_support/jobs/plugin_e2e.sh > $OUT/plugin_e2e.txt 2>&1
# N > 1 on one GPU: two ranks over gloo (torch coded / f64), and over the C ABI's gather with the tests' stand-in transport
make -s -C tests/cpu_harness
for extra in "" "--no-codec --gather torch" "--root-share 0.5"; do  # (no launcher: bench.py starts its two ranks itself)
  python bench.py --gpus 2 --same-device --backend gloo --rows 4000000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e $extra 2>/dev/null | tail -1 >> $OUT/bench_2rank_one_gpu.jsonl
done
for extra in "--gather abi" "--gather abi --root-share 0.5" ""; do
  STRSIM_RCCL_LIB=$(pwd)/tests/cpu_harness/libfake_rccl.so python bench.py --gpus 2 --same-device --backend gloo --rows 4000000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e $extra 2>/dev/null | tail -1 >> $OUT/bench_2rank_one_gpu.jsonl
done
cut -c1-400 $OUT/bench_2rank_one_gpu.jsonl
STEPS=10 WARMUP=5 bash bench_support/jobs/n8_first_contact.sh > $OUT/n8_first_contact_on_one_gpu.txt 2>&1
python tests/helpers/staging_child.py 32 4000000 > $OUT/staging_32_threads_default_budget.json 2>/dev/null
POLARS_STRSIM_STAGING_BUDGET_MB=1024 python tests/helpers/staging_child.py 32 4000000 > $OUT/staging_32_threads_1gib.json 2>/dev/null
# concurrent small calls: the C ABI alone, then the plugin ABI without and with the (opt-in) combiner
g++ -O2 -std=c++17 -pthread -Iinclude bench_support/micro/small_call_threads.cpp -o bench_support/micro/small_call_threads polars-strsim_amd/polars_strsim/libpolars_strsim_amd.so -Wl,-rpath,$PWD/polars-strsim_amd/polars_strsim
bench_support/micro/small_call_threads 3000 > $OUT/small_call_threads.txt 2>/dev/null
g++ -O2 -std=c++17 -pthread -Iinclude bench_support/micro/plugin_small_threads.cpp -o bench_support/micro/plugin_small_threads polars-strsim_amd/polars_strsim/libpolars_strsim_amd.so -Wl,-rpath,$PWD/polars-strsim_amd/polars_strsim
for mode in "POLARS_STRSIM_COALESCE=0" "POLARS_STRSIM_COALESCE=1"; do
  echo "== $mode" >> $OUT/plugin_small_threads.txt
  env $mode bench_support/micro/plugin_small_threads 2000 >> $OUT/plugin_small_threads.txt 2>/dev/null
done
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; grep -a "passed\|failed" $OUT/pytest_gpu.txt | tail -2
grep -h "k_lane\|k_wave" $OUT/prof_cfg2.txt | head -4; grep -h "traffic.json" $OUT/prof_cfg*.txt
