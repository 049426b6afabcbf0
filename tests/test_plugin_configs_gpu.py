"""BASELINE.json configs[0] ("cfg1") at its own workload, through the drop-in boundary: the 1 M-row frame of
bench_support/workload.py (seed 1, lengths U{0..16}, a-z) handed to `_polars_plugin_levenshtein` as Utf8View chunks -- what
Polars hands the reference's `levenshtein('a','b')` (reference polars_strsim/__init__.py:11-16, strsim.rs:41-107) -- and
EVERY row compared with the oracle, bit for bit.  Also the same frame as several chunks with 10 % nulls."""
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _arrow_column(off, val, validity=None):
    """u32 offsets + bytes -> pyarrow string array (zero-copy "u" layout; call_plugin casts it to the layout it sends)."""
    n = off.size - 1
    vbuf = None
    nulls = 0
    if validity is not None:
        vbuf = pa.py_buffer(np.packbits(validity, bitorder="little").tobytes())
        nulls = int(n - validity.sum())
    return pa.Array.from_buffers(pa.string(), n, [vbuf, pa.py_buffer(off.astype(np.int32).tobytes()), pa.py_buffer(val.tobytes())],
                                 null_count=nulls)


@pytest.fixture(scope="module")
def cfg1():
    from bench_support import workload as W
    measure, rows, law, lo, hi, seed = W.CONFIGS["cfg1"]
    assert (measure, rows, lo, hi, seed) == ("levenshtein", 1_000_000, 0, 16, 1)
    oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, rows)
    exp = O.batch(measure, oa, va, ob, vb, nthreads=8)
    return dict(rows=rows, cols=(oa, va, ob, vb), exp=exp)


@pytest.mark.parametrize("layout", ["vu", "u"])
@pytest.mark.parametrize("parallel", [False, True])
def test_cfg1_frame_through_the_plugin_abi(cfg1, layout, parallel):
    from strsim_amd import arrow_host as H
    oa, va, ob, vb = cfg1["cols"]
    got = H.call_plugin("levenshtein", _arrow_column(oa, va), _arrow_column(ob, vb), layout=layout, parallel=parallel)
    assert got.null_count == 0 and len(got) == cfg1["rows"]
    g = np.concatenate([c.to_numpy(zero_copy_only=False) for c in got.chunks])
    bad = np.nonzero(g.view(np.uint64) != cfg1["exp"].view(np.uint64))[0]
    assert bad.size == 0, (int(bad[0]), g[bad[0]], cfg1["exp"][bad[0]])


def test_cfg1_frame_in_chunks_with_nulls(cfg1):
    """10 % nulls in each column, chunk boundaries that do not line up: null in -> null out (README.md:69-70), every other
    row bit-exact (the bytes under a null slot stay in the values buffer, as in a Polars column)."""
    from strsim_amd import arrow_host as H
    oa, va, ob, vb = cfg1["cols"]
    n = cfg1["rows"]
    rng = np.random.default_rng(11)
    ka, kb = rng.random(n) >= 0.1, rng.random(n) >= 0.1
    a, b = _arrow_column(oa, va, ka), _arrow_column(ob, vb, kb)
    ca = pa.chunked_array([a[:333_333], a[333_333:333_334], a[333_334:]])
    cb = pa.chunked_array([b[:500_001], b[500_001:]])
    got = H.call_plugin("levenshtein", ca, cb)
    g = np.concatenate([c.to_numpy(zero_copy_only=False) for c in got.chunks])
    valid = np.concatenate([np.asarray(c.is_valid()) for c in got.chunks])
    assert np.array_equal(valid, ka & kb)
    bad = np.nonzero((g.view(np.uint64) != cfg1["exp"].view(np.uint64)) & valid)[0]
    assert bad.size == 0, (int(bad[0]), g[bad[0]], cfg1["exp"][bad[0]])
