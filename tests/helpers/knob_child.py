"""Child process of tests/test_knobs_gpu.py: one mixed frame through the plugin ABI in both engine modes (and a literal call) with
whatever knobs the parent put into the environment, every row against the oracle.  Prints KNOBS-OK on success."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "polars-strsim_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import pyarrow as pa

import gen
import oracle_lib as O
from strsim_amd import arrow_host as H

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
# short ASCII rows (the lane kernel), rows of 33..128 bytes, non-ASCII, a few long ones, empties; 3 % nulls in either column
A, B = gen.pairs(4001, rows * 80 // 100, gen.ASCII_LOWER, 0, 32)
A2, B2 = gen.pairs(4002, rows * 15 // 100, gen.ASCII_LOWER, 20, 128)
A3, B3 = gen.pairs(4003, rows - len(A) - len(A2) - 120, gen.MIXED, 0, 60)
A4, B4 = gen.pairs(4004, 120, gen.ASCII_LOWER, 200, 1500)  # (beyond 1 024 bytes: the pass that runs at retirement)
A, B = A + A2 + A3 + A4, B + B2 + B3 + B4
order = np.random.default_rng(7).permutation(len(A))
A, B = [A[i] for i in order], [B[i] for i in order]
An = [None if i % 37 == 5 else a for i, a in enumerate(A)]
Bn = [None if i % 41 == 7 else b for i, b in enumerate(B)]
# the engine hands chunks at odd boundaries
ca = pa.chunked_array([pa.array(An[:33_333], pa.string_view()), pa.array(An[33_333:], pa.string_view())])
cb = pa.chunked_array([pa.array(Bn[:50_001], pa.string_view()), pa.array(Bn[50_001:], pa.string_view())])


def check(measure, got, a_list, b_list):
    exp = O.batch_strings(measure, ["" if x is None else x for x in a_list], ["" if x is None else x for x in b_list], 8)
    g = got.combine_chunks() if hasattr(got, "combine_chunks") else got
    vals = g.to_numpy(zero_copy_only=False)
    valid = np.array([x is not None and y is not None for x, y in zip(a_list, b_list)])
    assert g.null_count == int((~valid).sum()), (measure, g.null_count)
    is_null = np.array(g.is_null().to_pylist())
    assert (is_null == ~valid).all(), measure
    bad = np.nonzero(valid & (np.asarray(vals, dtype=np.float64).view(np.uint64) != exp.view(np.uint64)))[0]
    assert bad.size == 0, (measure, int(bad[0]), a_list[bad[0]], b_list[bad[0]], vals[bad[0]], exp[bad[0]])


for parallel in (False, True):
    for measure in ("levenshtein", "jaro_winkler", "jaccard"):
        check(measure, H.call_plugin(measure, ca, cb, parallel=parallel), An, Bn)
lit = "phillipsburgh"
check("jaro", H.call_plugin("jaro", ca, [lit]), An, [lit] * len(An))
check("sorensen_dice", H.call_plugin("sorensen_dice", [lit], cb, parallel=True), [lit] * len(Bn), Bn)
print("KNOBS-OK", flush=True)
