#!/bin/bash
# end-to-end plugin-ABI timings (PCIe-inclusive) with the phase trace: bash bench_support/jobs/plugin_e2e.sh
export POLARS_STRSIM_TRACE=1
python - <<'PY'
import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyarrow as pa
from bench_support import workload as W
from strsim_amd import arrow_host as H
n = int(os.environ.get("E2E_ROWS", "10000000"))
_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, n)
def mk(o, v, valid=None):
    bufs = [None if valid is None else pa.py_buffer(np.packbits(valid, bitorder="little").tobytes()), pa.py_buffer(o.astype(np.int32)), pa.py_buffer(v)]
    return pa.Array.from_buffers(pa.string(), n, bufs, null_count=0 if valid is None else int(n - valid.sum())).cast(pa.string_view())
a, b = mk(oa, va), mk(ob, vb)
rng = np.random.default_rng(5)
an, bn = mk(oa, va, rng.random(n) >= 0.1), mk(ob, vb, rng.random(n) >= 0.1)
for name, x, y in (("no nulls", a, b), ("10% nulls", an, bn)):
    for par in (False, True):
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            r = H.call_plugin("levenshtein", x, y, parallel=par)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("== %s, engine_parallel=%s: best %.2f ms, %.3f G pairs/s, nulls out %d" % (name, par, best * 1e3, n / best / 1e9, r.null_count), flush=True)
PY
