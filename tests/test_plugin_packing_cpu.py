"""Host-side packing of the plugin layer (csrc/polars_plugin.cpp) without a GPU: every Arrow string layout, nulls,
multi-chunk inputs, sliced chunks, row sub-ranges and helper threads must produce the device layout
(u32 offsets rebased to 0 + packed UTF-8 bytes) that plain Python computes."""
import random

import numpy as np
import pyarrow as pa
import pytest

import gen
from strsim_amd import arrow_host as H


def expected(strings, r0, r1):
    sel = strings[r0:r1]
    bs = [(s or "").encode() if not isinstance(s, bytes) else s for s in sel]
    off = np.zeros(len(bs) + 1, dtype=np.uint32)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    val = np.frombuffer(b"".join(bs), dtype=np.uint8)
    valid = np.array([s is not None for s in sel], dtype=bool)
    return off, val, valid


def check(x, strings, layout, r0=0, r1=None, threads=1, null_bytes_kept=False):
    r1 = len(strings) if r1 is None else r1
    off, val, valid, rows = H.pack_series(x, layout=layout, r0=r0, r1=r1, threads=threads)
    assert rows == len(strings)
    eo, ev, evalid = expected(strings, r0, r1)
    assert (valid == evalid).all()
    if not null_bytes_kept:
        assert (off == eo).all() and (val == ev).all()
    else:  # offset layouts keep whatever bytes sit under a null slot: compare the valid rows only
        for i in np.nonzero(evalid)[0]:
            assert bytes(val[off[i]:off[i + 1]]) == bytes(ev[eo[i]:eo[i + 1]])


@pytest.mark.parametrize("layout", ["vu", "u", "U"])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_layouts_threads_and_ranges(layout, threads):
    A, _ = gen.pairs(5, 20000, gen.MIXED + gen.ASCII_LOWER, 0, 40)
    arr = pa.array(A, type=pa.string())
    check(arr, A, layout, threads=threads)
    check(arr, A, layout, r0=1234, r1=17777, threads=threads)
    check(arr, A, layout, r0=500, r1=500, threads=threads)


@pytest.mark.parametrize("layout", ["vu", "u"])
def test_nulls_chunks_and_slices(layout):
    rng = random.Random(3)
    A, _ = gen.pairs(6, 30000, gen.ASCII_LOWER, 0, 30)
    A = [None if rng.random() < 0.15 else s for s in A]
    arr = pa.array(A, type=pa.string())
    chunked = pa.chunked_array([arr[:1], arr[1:1], arr[1:9999], arr[9999:10000], arr[10000:]])
    check(chunked, A, layout, threads=4, null_bytes_kept=True)
    check(chunked, A, layout, r0=9990, r1=10010, threads=2, null_bytes_kept=True)
    sl = arr[777:20777]  # non-zero Arrow offset
    check(sl, A[777:20777], layout, threads=5, null_bytes_kept=True)
    allnull = pa.array([None] * 100, type=pa.string())
    off, val, valid, rows = H.pack_series(allnull, layout=layout)
    assert rows == 100 and not valid.any()


def test_view_inline_boundary_and_empty():
    S = ["", "a", "x" * 11, "y" * 12, "z" * 13, "é" * 6, "é" * 7, "w" * 300, ""]
    check(pa.array(S, type=pa.string()), S, "vu", threads=2)
    check(pa.array([], type=pa.string()), [], "vu")


def test_dtype_error_from_the_packer():
    with pytest.raises(H.PluginError, match="invalid series dtype"):
        H.pack_series(pa.array([1, 2, 3]))
