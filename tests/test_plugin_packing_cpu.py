"""Host-side packing of the plugin layer (csrc/polars_plugin.cpp) without a GPU: every Arrow string layout, nulls,
multi-chunk inputs, sliced chunks, row sub-ranges and helper threads must produce the device layout
(u32 offsets rebased to 0 + packed UTF-8 bytes) that plain Python computes."""
import random

import numpy as np
import pyarrow as pa
import pytest

import gen
import pack_harness as H


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


@pytest.mark.parametrize("threads", [1, 5])
def test_output_validity_is_the_and_of_the_inputs(threads):
    """The word-wise validity builder against plain Python: chunk boundaries and Arrow offsets that are not multiples of 8 or
    64, one side without nulls, a literal side, a null literal, n not a multiple of 64; null slots get 0.0 in the values."""
    rng = random.Random(4)
    for n in (1, 63, 64, 65, 1000, 300_000 + 37):
        A = [None if rng.random() < 0.15 else "x" for _ in range(n)]
        B = [None if rng.random() < 0.05 else "y" for _ in range(n)]
        pa_a, pa_b = pa.array(A, type=pa.string()), pa.array(B, type=pa.string())
        cuts = sorted({0, n, *(rng.randrange(n + 1) for _ in range(4))})
        ca = pa.chunked_array([pa_a[i:j] for i, j in zip(cuts[:-1], cuts[1:])] or [pa_a])
        cutsb = sorted({0, n, *(rng.randrange(n + 1) for _ in range(2))})
        cb = pa.chunked_array([pa_b[i:j] for i, j in zip(cutsb[:-1], cutsb[1:])] or [pa_b])
        exp = np.array([x is not None and y is not None for x, y in zip(A, B)])
        vals = np.full(n, 7.0)
        got, nulls = H.validity(ca, cb, threads=threads, vals=vals)
        assert (got == exp).all() and nulls == int((~exp).sum()), n
        assert (vals[~exp] == 0.0).all() and (vals[exp] == 7.0).all()
        for layouts in (("u", "vu"), ("U", "u")):
            got, nulls = H.validity(ca, cb, layouts=layouts, threads=threads)
            assert (got == exp).all() and nulls == int((~exp).sum())
        # one side without nulls; a literal on either side; a null literal
        got, nulls = H.validity(ca, ["z"] * n, threads=threads)
        assert (got == np.array([x is not None for x in A])).all()
        if n > 1:
            got, nulls = H.validity(ca, "lit", threads=threads)
            assert (got == np.array([x is not None for x in A])).all() and nulls == sum(x is None for x in A)
            got, nulls = H.validity("lit", cb, threads=threads)
            assert (got == np.array([y is not None for y in B])).all()
            got, nulls = H.validity(ca, None, threads=threads)
            assert not got.any() and nulls == n


@pytest.mark.parametrize("threads", [1, 4, 8])
def test_view_native_packer(threads):
    """pack_slice_views: the views leave as they lie, a null slot as the empty string, a string beyond 12 bytes in the
    thread's segment of the long-string area with its offset in the view -- sized exactly, from an estimate that holds, and from
    one that does not (the packer then sizes exactly itself); chunks at odd offsets, sliced arrays."""
    rng = random.Random(13)
    A, _ = gen.pairs(56, 70_000, gen.ASCII_LOWER + gen.MIXED, 0, 40)
    A += ["x" * 13, "y" * 12, "", "z" * 300, "w" * 5000]
    A = [None if rng.random() < 0.05 else x for x in A]
    arr = pa.array(A, type=pa.string())
    x = pa.chunked_array([arr[:20_001], arr[20_001:20_001], arr[20_001:]])
    want = [(s or "").encode() for s in A]
    for r0, r1 in ((0, len(A)), (1234, 66_000), (500, 500)):
        for est in (None, 64, 0.01):
            got, total = H.pack_views(x, r0, r1, threads=threads, long_bytes_per_row=est)
            assert got == want[r0:r1], (r0, r1, est)
            assert total == sum(len(b) for b in want[r0:r1])
    sl = arr[777:50_777]  # non-zero Arrow offset
    got, _ = H.pack_views(sl, 0, 50_000, threads=threads)
    assert got == want[777:50_777]


@pytest.mark.parametrize("threads", [1, 4, 8])
def test_one_pass_packer(threads):
    """pack_slice_onepass: per-thread segments sized from a bytes-per-row budget, one length byte per row, no size pass -- equal
    to the plain-Python packing when the budget holds, a clean "no" when a segment overflows or a string exceeds 255 bytes."""
    rng = random.Random(12)
    A, _ = gen.pairs(55, 70_000, gen.ASCII_LOWER, 0, 40)  # > 12 bytes: out-of-line views too
    A = [None if rng.random() < 0.05 else x for x in A]
    x = pa.chunked_array([pa.array(A[:20_001], type=pa.string()), pa.array(A[20_001:], type=pa.string())])
    for r0, r1 in ((0, len(A)), (1234, 66_000)):
        eo, ev, _ = expected(A, r0, r1)
        got = H.pack_onepass(x, r0, r1, bytes_per_row=24, threads=threads)
        assert got is not None
        lens, val, nseg = got
        assert nseg == min(threads, max((r1 - r0) // 16384, 1))
        assert (lens == np.diff(eo).astype(np.uint8)).all() and (val == ev).all()
    assert H.pack_onepass(x, 0, len(A), bytes_per_row=2, threads=threads) is None        # the budget does not hold
    B = list(A)
    B[40_000] = "q" * 300
    assert H.pack_onepass(pa.array(B, type=pa.string()), 0, len(B), bytes_per_row=24, threads=threads) is None  # > 255 bytes


def test_engine_parallel_calls_borrow_helpers_from_one_bounded_budget():
    """CallerContext PARALLEL (strsim.rs:53): the reference packs on the calling thread alone so as not to oversubscribe the CPUs.
    Here such calls may borrow helper threads -- from ONE process-wide budget of half the CPU quota, whatever the order the
    engine's threads arrive in (sizing each call from a one-shot read of a counter of calls in flight handed sixteen
    staggered calls about 70 helpers)."""
    import os
    quota = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, min(quota, -(-int(q) // int(per))))
    except Exception:
        pass
    budget = min(quota, 32) // 2
    threads, lent, after = H.pack_grants(True, 16, 10_000_000)
    assert after == 0                                   # everything borrowed came back
    assert lent == sum(t - 1 for t in threads) <= budget
    assert threads[0] == 1 + budget or budget == 0      # a lone call gets the whole budget ...
    assert all(t == 1 for t in threads[1:])             # ... and the calls that arrive while it runs pack on their own thread
    # small calls never take helpers; a sequential engine context is not rationed
    assert H.pack_grants(True, 3, 20_000) == ([1, 1, 1], 0, 0)
    seq, lent, after = H.pack_grants(False, 3, 10_000_000)
    assert lent == 0 and after == 0 and all(t == min(quota, 32) for t in seq)


def test_engine_parallel_strict_rule_by_environment(monkeypatch):
    """POLARS_STRSIM_PARALLEL_PACK=0 (read once per process: a fresh interpreter) keeps the reference's rule to the letter."""
    import subprocess, sys, os
    code = ("import sys; sys.path[:0] = [%r, %r]; import pack_harness as P; print(P.pack_grants(True, 2, 10_000_000))"
            % (os.path.join(H.ROOT, "polars-strsim_amd"), os.path.join(H.ROOT, "tests")))
    env = dict(os.environ, POLARS_STRSIM_PARALLEL_PACK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout.strip().endswith("([1, 1], 0, 0)")
    env = dict(os.environ, POLARS_STRSIM_PACK_THREADS="3")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    threads, lent, after = eval(r.stdout.strip().splitlines()[-1])
    assert after == 0 and all(1 <= t <= 3 for t in threads) and lent == sum(t - 1 for t in threads)  # the cap holds per call
