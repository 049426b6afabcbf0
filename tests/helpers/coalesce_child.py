"""Child process of tests/test_plugin_coalesce_gpu.py: many engine threads, many SMALL calls each through the plugin ABI.

  coalesce_child.py <threads> <calls per thread>

POLARS_STRSIM_COALESCE* come from the parent's environment.  Every call has its own frame (1 .. 3000 rows: short ASCII rows, and
in some frames rows for the kernels behind the first one -- 33..128 bytes, non-ASCII, a long one -- and nulls) and its own measure;
every result is compared with the oracle bit for bit.  Prints one JSON line with the combiner's counters."""
import ctypes as C
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "polars-strsim_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import pyarrow as pa

import gen
import oracle_lib as O
import strsim_amd
from strsim_amd import arrow_host as H

threads, calls = int(sys.argv[1]), int(sys.argv[2])
MEASURES = list(O.MEASURES)
rng = np.random.default_rng(11)
frames = []
for k in range(24):  # a pool of frames; every call picks one
    n = int(rng.integers(1, 3000))
    A, B = gen.pairs(7000 + k, n, gen.ASCII_LOWER, 0, 32)
    if k % 3 == 0:  # rows the first kernel leaves behind
        m = max(1, n // 20)
        A2, B2 = gen.pairs(7100 + k, m, gen.ASCII_LOWER, 33, 128)
        A3, B3 = gen.pairs(7200 + k, m, gen.MIXED, 0, 40)
        A, B = A + A2 + A3 + ["x" * 1500], B + B2 + B3 + ["x" * 700 + "y" * 700]
    An = [None if (k % 4 == 1 and i % 17 == 3) else a for i, a in enumerate(A)]
    Bn = [None if (k % 4 == 1 and i % 23 == 5) else b for i, b in enumerate(B)]
    fa, fb = pa.array(An, pa.string_view()), pa.array(Bn, pa.string_view())
    exp = {m: O.batch_strings(m, ["" if x is None else x for x in An], ["" if x is None else x for x in Bn], 2) for m in MEASURES}
    valid = np.array([x is not None and y is not None for x, y in zip(An, Bn)])
    frames.append((fa, fb, exp, valid))

H.call_plugin("levenshtein", frames[0][0], frames[0][1])  # library + context warm-up
bad = []


def work(t):
    r = np.random.default_rng(100 + t)
    for c in range(calls):
        fa, fb, exp, valid = frames[int(r.integers(0, len(frames)))]
        m = MEASURES[int(r.integers(0, 5))]
        got = H.call_plugin(m, fa, fb, parallel=True)
        g = got.combine_chunks() if hasattr(got, "combine_chunks") else got
        vals = np.asarray(g.to_numpy(zero_copy_only=False), dtype=np.float64)
        nulls = np.array(g.is_null().to_pylist())
        if len(g) != len(valid) or (nulls != ~valid).any() or (valid & (vals.view(np.uint64) != exp[m].view(np.uint64))).any():
            bad.append((t, c, m, len(g)))
            return


ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
for t in ts:
    t.start()
for t in ts:
    t.join()
out = (C.c_uint64 * 4)()
strsim_amd.lib()._polars_plugin_strsim_coalesce_stats(out)
print(json.dumps({"threads": threads, "calls": calls, "bad": bad, "combined_launches": int(out[0]), "calls_combined": int(out[1]),
                  "most_calls_in_one_launch": int(out[2]), "calls_on_the_ordinary_path": int(out[3])}), flush=True)
