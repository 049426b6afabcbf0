#!/bin/bash
# round 6: does the new cold-call measurement (two 1 s sleeps before the preheat) disturb the headline? + cfg3 after the KEEP_EQ fix + the GPU suite
OUT=gpurun_out/r6c; mkdir -p $OUT
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 >> $OUT/cfg2_default.jsonl
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-modes 2>/dev/null | tail -1 >> $OUT/cfg2_noextra.jsonl
done
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 >> $OUT/cfg3.jsonl
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 >> $OUT/cfg3.jsonl
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c/*.jsonl")):
    for l in open(f):
        d = json.loads(l); r = d["roofline"]
        print("%-28s %9.1f M/s %8.4f ms/step kernel %.4f + %.4f ms frac %.4f default_mode %s cold %s idle %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], r["kernel_ms"], r["wave_kernel_ms"], r["frac"], d.get("value_default_mode"), d.get("cold_first_call_ms"), d.get("idle_gpu_call_ms")))
PY
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
