#!/usr/bin/env python3
"""End-to-end rate THROUGH the Polars plugin ABI (host Arrow buffers in, host f64 out): view compaction on the
host, H2D over PCIe, kernels, D2H.  This is never bench.py's `value` (which is device-resident); it is the
PCIe-inclusive figure DESIGN.md quotes.  Usage: python bench_support/bench_plugin_e2e.py [rows] [layout]"""
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


def column(off, val, layout):
    n = len(off) - 1
    arr = pa.StringArray.from_buffers(n, pa.py_buffer(off.astype(np.int32)), pa.py_buffer(val))
    return arr.cast(pa.string_view()) if layout == "vu" else arr


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    layout = sys.argv[2] if len(sys.argv) > 2 else "vu"
    _, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
    oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, rows)
    a, b = column(oa, va, layout), column(ob, vb, layout)
    for name in ("levenshtein", "jaro_winkler"):
        H.call_plugin(name, a[:1000], b[:1000], layout=layout)  # context + library warm-up
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            out = H.call_plugin(name, a, b, layout=layout)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        bytes_in = len(va) + len(vb) + 8 * (rows + 1)
        print(f"{name}: {rows} rows through _polars_plugin_{name} ({layout}): {best*1e3:.1f} ms best of 3 = "
              f"{rows/best/1e6:.1f} M pairs/s end to end; {bytes_in/best/1e9:.2f} GB/s of input, {len(out)} rows out")


if __name__ == "__main__":
    main()
