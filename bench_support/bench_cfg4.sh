python bench.py --config cfg4 --rows 100000000 --steps 10 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('cfg4 100M rows: value %.1f M rows/s  ms/step %.3f lane_ms %.3f wave_ms %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['wave_kernel_ms']))"
