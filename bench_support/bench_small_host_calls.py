"""Latency of strsim_pairs_host (thin C ABI, host buffers in, host f64 out) for small calls, through ctypes."""
import sys
import time

sys.path.insert(0, "polars-strsim_amd")
sys.path.insert(0, ".")
import numpy as np
import strsim_amd as S
from bench_support import workload as W

_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
sizes = [int(x) for x in sys.argv[1:]] or [1, 100, 1000, 10_000, 100_000, 200_000]
with S.Context(0) as ctx:
    for rows in sizes:
        oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, rows)
        ctx.pairs_host("levenshtein", oa, va, ob, vb)
        ts = []
        for _ in range(200 if rows <= 10_000 else 30):
            t0 = time.perf_counter()
            ctx.pairs_host("levenshtein", oa, va, ob, vb)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"rows={rows:>8}: median {ts[len(ts)//2]*1e6:9.1f} us, min {ts[0]*1e6:9.1f} us ({rows/ts[len(ts)//2]/1e6:8.2f} M pairs/s)")
