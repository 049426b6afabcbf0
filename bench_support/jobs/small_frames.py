"""Lane-kernel time of small and mid-size frames (cfg1's length law) -- launch sizing and one-launch calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd")); sys.path.insert(0, ROOT)
import torch
import strsim_amd as S
from bench_support import workload as W
dev = torch.device("cuda", 0)
m, _, law, lo, hi, seed = W.CONFIGS["cfg1"]
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream, one_launch=True)
for rows in [int(x) for x in os.environ.get('ROWS', '1000,10000,100000,300000,1000000,3000000,10000000').split(',')]:
    oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, 0, rows, dev)
    out = torch.empty(rows, dtype=torch.float64, device=dev)
    for _ in range(20):
        ctx.pairs_device(m, oa, va, ob, vb, out=out)
    ctx.synchronize()
    n = 200
    b = ctx.enqueued_ops
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.pairs_device(m, oa, va, ob, vb, out=out)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("rows %9d: %8.2f us per call, %7.2f G pairs/s, %.1f ops per call, late %d" % (rows, dt * 1e6, rows / dt / 1e9, (ctx.enqueued_ops - b) / n, ctx.last_late_rows))
