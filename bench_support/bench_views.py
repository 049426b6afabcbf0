"""the plugin ABI on view columns, view-native slices against packed ones, both engine modes, cfg1 and cfg2 lengths.
   python bench_support/bench_views.py [rows]      (PCIe-inclusive wall time per call, POLARS_STRSIM_TRACE-style split on stderr)"""
import os
import sys
import time

import numpy as np
import pyarrow as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
from bench_support import workload as W  # noqa: E402
from strsim_amd import arrow_host as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for cfg, (lo, hi, seed) in (("cfg1 lengths U{0..16}", (0, 16, 1)), ("cfg2 lengths U{1..32}", (1, 32, 2))):
    oa, va, ob, vb = W.host_columns(seed, W.UNIFORM, lo, hi, 0, n)
    a = pa.StringArray.from_buffers(n, pa.py_buffer(oa.astype(np.int32)), pa.py_buffer(va)).cast(pa.string_view())
    b = pa.StringArray.from_buffers(n, pa.py_buffer(ob.astype(np.int32)), pa.py_buffer(vb)).cast(pa.string_view())
    inline = float(np.mean(np.diff(oa.astype(np.int64)) <= 12))
    ref = None
    for parallel in (False, True):
        for views in ("0", "1"):
            os.environ["POLARS_STRSIM_VIEWS"] = views
            ts = []
            for rep in range(5):
                t0 = time.perf_counter()
                out = H.call_plugin("levenshtein", a, b, layout="vu", parallel=parallel)
                ts.append(time.perf_counter() - t0)
            got = out.to_numpy(zero_copy_only=False)
            if ref is None:
                ref = got
            assert (got.view(np.uint64) == ref.view(np.uint64)).all()
            best = min(ts[1:])
            print("%-22s %4.0f %% inline  engine-parallel=%d views=%s : %7.2f ms per %d rows  (%.2f G pairs/s)"
                  % (cfg, 100 * inline, parallel, views, best * 1e3, n, n / best / 1e9), flush=True)
