#!/usr/bin/env python3
"""Per-call latency of the plugin ABI for small morsels (Polars calls elementwise plugins per batch/group)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import pyarrow as pa

from bench_support import workload as W
from strsim_amd import arrow_host as H

_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
SIZES = [int(x) for x in sys.argv[1:]] or [1, 100, 10_000, 100_000, 1_000_000]
for rows in SIZES:
    oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, rows)
    a = pa.StringArray.from_buffers(rows, pa.py_buffer(oa.astype(np.int32)), pa.py_buffer(va)).cast(pa.string_view())
    b = pa.StringArray.from_buffers(rows, pa.py_buffer(ob.astype(np.int32)), pa.py_buffer(vb)).cast(pa.string_view())
    t0 = time.perf_counter()
    H.call_plugin("levenshtein", a, b)
    first = time.perf_counter() - t0
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        H.call_plugin("levenshtein", a, b)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"rows={rows:>8}: first call {first*1e3:8.2f} ms, median {ts[len(ts)//2]*1e6:9.1f} us, min {ts[0]*1e6:9.1f} us "
          f"({rows/ts[len(ts)//2]/1e6:8.2f} M pairs/s)")
