set -x
mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_parity.py tests/test_plugin_abi_gpu.py -x -q -m gpu > gpurun_out/r3b/pytest_parity.txt 2>&1; tail -5 gpurun_out/r3b/pytest_parity.txt
for m in levenshtein jaro jaro_winkler jaccard sorensen_dice; do
  python bench.py --warmup 5 --steps 20 --measure $m --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['value']), 'Mpairs/s', d['ms_per_step'], 'ms/step kernel', d['roofline']['kernel_ms'], 'frac', round(d['roofline']['frac'],4))"
done
python bench.py --config cfg3 --warmup 3 --steps 10 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | cut -c1-400
