"""Child process of tests/test_plugin_staging_gpu.py: many engine threads, one large call each, through the plugin ABI.

  staging_child.py <threads> <rows per call>

POLARS_STRSIM_STAGING_BUDGET_MB comes from the parent's environment.  Every thread's result is compared with the oracle bit for
bit; afterwards -- every thread idle -- the process-wide staging (pinned + device) must be within the budget.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "polars-strsim_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import pyarrow as pa

import oracle_lib as O
import strsim_amd
from bench_support import workload as W
from strsim_amd import arrow_host as H

threads, rows = int(sys.argv[1]), int(sys.argv[2])
_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, rows)
mk = lambda o, v: pa.StringArray.from_buffers(rows, pa.py_buffer(o.astype(np.int32)), pa.py_buffer(v)).cast(pa.string_view())
a, b = mk(oa, va), mk(ob, vb)
exp = {m: O.batch(m, oa, va, ob, vb, nthreads=8) for m in ("levenshtein", "jaccard")}


def stats():
    out = (C.c_uint64 * 8)()
    strsim_amd.lib()._polars_plugin_strsim_staging_stats(out)
    keys = ("live_pinned", "live_device", "budget", "sets", "sets_in_use", "sets_released", "calls_waited", "peak_live_at_a_return")
    return dict(zip(keys, [int(v) for v in out]))


H.call_plugin("levenshtein", a[:1000], b[:1000])  # library + context warm-up
bad, times, peak = [], [], [0]


def work(i):
    m = ("levenshtein", "jaccard")[i % 2]
    t0 = time.perf_counter()
    got = H.call_plugin(m, a, b, parallel=True)  # (engine threads call in the engine-parallel mode)
    times.append(time.perf_counter() - t0)
    vals = got.combine_chunks().to_numpy(zero_copy_only=False) if hasattr(got, "combine_chunks") else got.to_numpy(zero_copy_only=False)
    n_bad = int((np.asarray(vals, dtype=np.float64).view(np.uint64) != exp[m].view(np.uint64)).sum())
    if n_bad:
        bad.append((i, m, n_bad))


def watch(stop):
    while not stop.is_set():
        s = stats()
        peak[0] = max(peak[0], s["live_pinned"] + s["live_device"])
        time.sleep(0.002)


stop = threading.Event()
w = threading.Thread(target=watch, args=(stop,))
w.start()
t0 = time.perf_counter()
ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
for t in ts:
    t.start()
for t in ts:
    t.join()
wall = time.perf_counter() - t0
stop.set()
w.join()
idle = stats()
# a lone call afterwards still works, and leaves the process within the budget again
lone = H.call_plugin("levenshtein", a, b)
lv = lone.combine_chunks().to_numpy(zero_copy_only=False) if hasattr(lone, "combine_chunks") else lone.to_numpy(zero_copy_only=False)
lone_bad = int((np.asarray(lv, dtype=np.float64).view(np.uint64) != exp["levenshtein"].view(np.uint64)).sum())
print(json.dumps({"threads": threads, "rows": rows, "bad": bad, "lone_bad": lone_bad, "wall_s": wall, "idle": idle, "after_lone": stats(),
                  "peak_live_sampled": peak[0], "slowest_call_s": max(times), "fastest_call_s": min(times)}), flush=True)
