#!/bin/bash
# Round-5 evidence run (one gpurun call): counter profiles of the final build first (they write profiles/traffic.json, which the
# bench lines taken afterwards replay), then the bench lines of every config, the measures, the side benches, the GPU suite.
OUT=gpurun_out/r5_final; rm -rf $OUT; mkdir -p $OUT
STEPS=150 WARMUP=20 TRAFFIC_KEY=cfg2:levenshtein:100000000 TRAFFIC_KERNEL=k_lane_stage bash bench_support/profile.sh r5_cfg2 > $OUT/prof_cfg2.txt 2>&1
STEPS=8 WARMUP=3 TRAFFIC_KEY=cfg3:jaro_winkler:100000000 TRAFFIC_KERNEL=k_lane_stage+k_lane_wide bash bench_support/profile.sh r5_cfg3 --config cfg3 > $OUT/prof_cfg3.txt 2>&1
STEPS=3 WARMUP=1 TRAFFIC_KEY=cfg5:levenshtein:10000000 TRAFFIC_KERNEL=k_lane_stage+k_wave_pairs bash bench_support/profile.sh r5_cfg5 --config cfg5 > $OUT/prof_cfg5.txt 2>&1
cp profiles/traffic.json $OUT/traffic.json
python bench.py > $OUT/bench_default.jsonl 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.jsonl | cut -c1-600
tail -1 $OUT/bench_default.jsonl > $OUT/bench_lines.jsonl
for args in "--config cfg1 --steps 200" "--config cfg3 --steps 10 --warmup 3" "--config cfg5 --steps 3 --warmup 1" "--measure all --rows 100000000 --steps 10 --warmup 3" "--measure jaro" "--measure jaro_winkler" "--measure jaccard" "--measure sorensen_dice"; do
  python bench.py $args --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 >> $OUT/bench_lines.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r5_final/bench_lines.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print("%-64s %9.1f M/s %8.4f ms/step kernel %.4f + %.4f ms frac %.4f traffic %s ops/step %s %s" % (d["metric"][:64], d["value"], d["ms_per_step"], r["kernel_ms"], r["wave_kernel_ms"], r["frac"], r["traffic"], d["config"]["enqueued_kernels_and_copies_per_step"], d.get("gcups", "")))
PY
python bench_support/jobs/small_frames.py > $OUT/small_frames.txt 2>/dev/null
python bench_support/bench_literal.py > $OUT/literal.txt 2>/dev/null
python bench_support/bench_mid_ascii.py 2>/dev/null | tail -5 > $OUT/mid_ascii.txt
bash bench_support/jobs/plugin_e2e.sh > $OUT/plugin_e2e.txt 2>&1
for extra in "" "--no-codec" "--root-share 0.5"; do  # (no launcher: bench.py starts its two ranks itself)
  python bench.py --gpus 2 --same-device --backend gloo --rows 4000000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e $extra 2>/dev/null | tail -1 >> $OUT/bench_2rank_gloo_one_gpu.jsonl
done
python bench_support/bench_root_rehearsal.py 2>/dev/null | grep -v amdgpu.ids > $OUT/root_rehearsal.txt
cut -c1-300 $OUT/bench_2rank_gloo_one_gpu.jsonl
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -2 $OUT/pytest_gpu.txt
grep -h "k_lane\|k_wave" $OUT/prof_cfg2.txt | head -4; grep -h "traffic.json" $OUT/prof_cfg*.txt
