#!/usr/bin/env python3
"""Host-side packing speed of the plugin (Utf8View -> offsets + values), no GPU: rows/s of _strsim_test_pack_series."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import pyarrow as pa

from bench_support import workload as W
sys.path.insert(0, os.path.join(ROOT, "tests"))
from strsim_amd import arrow_host as H
import pack_harness

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
oa, va, _, _ = W.host_columns(2, W.UNIFORM, 1, 32, 0, n)
a = pa.StringArray.from_buffers(n, pa.py_buffer(oa.astype(np.int32)), pa.py_buffer(va)).cast(pa.string_view())
fn = pack_harness.lib()._strsim_test_pack_series
chunks, dtype = H._chunks(a, "vu")
off = np.zeros(n + 1, dtype=np.uint32)
val = np.zeros(len(va) + 64 * 1024, dtype=np.uint8)
valid = np.zeros(n + 1, dtype=np.uint8)
best = None
for _ in range(5):
    ex = H._Exported("col", chunks, dtype)
    se = H.SeriesExport()
    ex.fill(se)
    rows, used = C.c_uint64(), C.c_uint64()
    t0 = time.perf_counter()
    rc = fn(C.byref(se), 0, n, off.ctypes.data, val.ctypes.data, val.size, C.byref(rows), C.byref(used), None, threads)
    dt = time.perf_counter() - t0
    assert rc == 0
    best = dt if best is None else min(best, dt)
ok = np.array_equal(off, oa) and np.array_equal(val[: used.value], va)
print(f"pack {n} rows, {threads} threads: {best*1e3:.2f} ms = {n/best/1e6:.1f} M rows/s (output {'matches' if ok else 'DIFFERS'})")
