#!/bin/bash
# Round-4 sanity run after a code change (one gpurun call): GPU test suite, then the bench lines of cfg2 / cfg3 / cfg5.
OUT=gpurun_out/r4_check; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
: > $OUT/bench_lines.jsonl
for args in "" "--config cfg3 --steps 10 --warmup 3" "--config cfg5 --steps 3 --warmup 1" "--measure jaro_winkler" "--measure jaccard"; do
  python bench.py $args --no-cpu-baseline --no-e2e 2>>$OUT/bench.err | tail -1 >> $OUT/bench_lines.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r4_check/bench_lines.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print("%-70s %9.1f M/s  %8.4f ms/step  kernel %.4f + %.4f ms  frac %.4f  ops/step %s" % (d["metric"][:70], d["value"], d["ms_per_step"], r["kernel_ms"], r["wave_kernel_ms"], r["frac"], d["config"]["enqueued_kernels_and_copies_per_step"]))
PY
POLARS_STRSIM_VIEWS=0 python bench_support/bench_views.py 10000000 2>/dev/null | grep "views=0" | tee $OUT/engine_parallel.txt
