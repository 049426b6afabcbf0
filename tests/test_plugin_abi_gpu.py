"""The Polars plugin ABI end to end on the GPU, with pyarrow standing in for the Polars engine.

Covers what the reference leaves untested: Utf8View vs offset layouts, multi-chunk inputs with
misaligned chunk boundaries, sliced arrays (non-zero Arrow offset), nulls in either/both columns, literals on
either side, the README demo table.  Oracle = oracle/ (bit-exact)."""
import numpy as np
import pyarrow as pa
import pytest

import gen
import oracle_lib as O
from golden_data import readme_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from strsim_amd import arrow_host
    return arrow_host


def expect(measure, A, B):
    """oracle with null propagation and literal broadcast -> list of float or None"""
    n = max(len(A), len(B))
    a = A if len(A) > 1 or n == 1 else A * n
    b = B if len(B) > 1 or n == 1 else B * n
    out = []
    for x, y in zip(a, b):
        out.append(None if x is None or y is None else O.pair(measure, x, y))
    return out


def check(got, exp):
    got = got.to_pylist()
    assert len(got) == len(exp)
    for i, (g, e) in enumerate(zip(got, exp)):
        if e is None:
            assert g is None, i
        else:
            assert g is not None and np.float64(g).view(np.uint64) == np.float64(e).view(np.uint64), (i, g, e)


def test_readme_demo_table(H):
    rows = readme_table()
    A = [r["name_a"] for r in rows]
    B = [r["name_b"] for r in rows]
    for m in O.MEASURES:
        probe = {}
        got = H.call_plugin(m, A, B, names=("name_a", "name_b"), _probe=probe).to_pylist()
        assert probe["series_released"] == [1, 1] and probe["arrays_released"] == [True, True]
        assert probe["name"] == "name_a"
        for r, g in zip(rows, got):
            if r[m] is None:
                assert g is None  # README.md:69-70
            else:
                assert abs(g - r[m]) < 5e-7


@pytest.mark.parametrize("layout", ["vu", "u", "U", ("vu", "u")])
@pytest.mark.parametrize("measure", O.MEASURES)
def test_layouts_long_and_short_strings(H, layout, measure):
    A, B = gen.pairs(21, 3000, gen.ASCII_LOWER, 0, 40)  # > 12 bytes => out-of-line views
    A2, B2 = gen.pairs(22, 300, gen.MIXED, 0, 30)
    A, B = A + A2, B + B2
    check(H.call_plugin(measure, A, B, layout=layout), expect(measure, A, B))


@pytest.mark.parametrize("measure", ["levenshtein", "jaro_winkler", "jaccard"])
def test_nulls_chunks_and_slices(H, measure):
    import random
    rng = random.Random(9)
    A, B = gen.pairs(23, 5000, gen.ASCII_LOWER, 0, 24)
    A = [None if rng.random() < 0.1 else x for x in A]
    B = [None if rng.random() < 0.1 else x for x in B]
    pa_a = pa.array(A, type=pa.string())
    pa_b = pa.array(B, type=pa.string())
    # misaligned chunk boundaries + sliced chunks (non-zero offset)
    ca = pa.chunked_array([pa_a[:7], pa_a[7:1000], pa_a[1000:1000], pa_a[1000:4999], pa_a[4999:]])
    cb = pa.chunked_array([pa_b[:2048], pa_b[2048:2049], pa_b[2049:]])
    for layout in ("vu", "u"):
        check(H.call_plugin(measure, ca, cb, layout=layout), expect(measure, A, B))
    # a slice of a bigger array on both sides
    check(H.call_plugin(measure, pa_a[100:3100], pa_b[100:3100]), expect(measure, A[100:3100], B[100:3100]))


@pytest.mark.parametrize("measure", O.MEASURES)
def test_literal_either_side_and_null_cases(H, measure):
    A, _ = gen.pairs(24, 2000, gen.ASCII_LOWER, 0, 20)
    A[5] = None
    check(H.call_plugin(measure, A, "phillips"), expect(measure, A, ["phillips"]))     # strsim.rs:61-63
    check(H.call_plugin(measure, "phillips", A), expect(measure, ["phillips"], A))     # strsim.rs:64-66 (intended semantics)
    check(H.call_plugin(measure, A, [None]), [None] * len(A))                           # null literal: all-null column
    check(H.call_plugin(measure, [None] * 10, [None] * 10), [None] * 10)               # all-null inputs
    check(H.call_plugin(measure, ["x"], ["x"]), [1.0])
    assert H.call_plugin(measure, [], []).to_pylist() == []


@pytest.mark.parametrize("measure", O.MEASURES)
def test_small_calls_computed_in_place(H, measure):
    """Calls of <= 131 072 rows and <= 2 MiB per column: the kernels read the pinned staging and write the pinned result
    buffer (no copy engine).  Every row class goes through it -- lane, wide, non-ASCII, wave, > 1024-byte strings (second
    pass) -- plus literals; then the same frames with the path switched off."""
    A, B = gen.pairs(31, 600, gen.ASCII_LOWER, 0, 32)
    A2, B2 = gen.pairs(32, 100, gen.MIXED, 0, 200)
    A3, B3 = gen.pairs(33, 40, gen.ASCII_LOWER, 100, 900)
    A = ["x" * 1500, "\u00e9" * 700] + A + A2 + A3
    B = ["x" * 700 + "y" * 700, "e" * 1300] + B + B2 + B3
    exp = expect(measure, A, B)
    for n in (1, 2, 3, 63, 64, 65, len(A)):
        check(H.call_plugin(measure, A[:n], B[:n]), exp[:n])
    check(H.call_plugin(measure, A, "x" * 1200), expect(measure, A, ["x" * 1200]))
    check(H.call_plugin(measure, "phillips", B), expect(measure, ["phillips"], B))
    k = -(-4097 // len(A))
    An, Bn = (A * k)[:4097], (B * k)[:4097]
    check(H.call_plugin(measure, An, Bn), (exp * k)[:4097])
    # the same frames through the copy path (H2D, device buffers, D2H): the limit is read per call
    import os
    os.environ["POLARS_STRSIM_DIRECT_ROWS"] = "0"
    try:
        check(H.call_plugin(measure, A, B), exp)
        check(H.call_plugin(measure, A[:3], B[:3]), exp[:3])
        check(H.call_plugin(measure, A, "x" * 1200), expect(measure, A, ["x" * 1200]))
    finally:
        del os.environ["POLARS_STRSIM_DIRECT_ROWS"]


def test_threaded_packing_and_multi_slice_pipeline(H):
    """> 2 M rows: several pipeline slices, each packed by helper threads (views with nulls, out-of-line strings)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_support import workload as W
    n = 4_700_000
    oa, va, ob, vb = W.host_columns(11, W.UNIFORM, 0, 40, 0, n)
    a = pa.StringArray.from_buffers(n, pa.py_buffer(oa.astype(np.int32)), pa.py_buffer(va))
    b = pa.StringArray.from_buffers(n, pa.py_buffer(ob.astype(np.int32)), pa.py_buffer(vb))
    # nulls in a: every 1000th row
    mask = np.zeros(n, dtype=bool)
    mask[::1000] = True
    a = pa.array(a.to_numpy(zero_copy_only=False), type=pa.string(), mask=mask) if False else pa.StringArray.from_buffers(
        n, pa.py_buffer(oa.astype(np.int32)), pa.py_buffer(va), pa.py_buffer(np.packbits(~mask, bitorder="little")), int(mask.sum()))
    exp = O.batch("levenshtein", oa, va, ob, vb, nthreads=8)
    for layout in ("vu", "u"):
        got = H.call_plugin("levenshtein", pa.chunked_array([a[:1_000_003], a[1_000_003:]]), b, layout=layout)
        g = got.to_numpy(zero_copy_only=False)
        assert got.null_count == int(mask.sum())
        ok = ~mask
        assert np.isnan(g[mask]).all()
        assert (g[ok].view(np.uint64) == exp[ok].view(np.uint64)).all()


def test_concurrent_calls_from_several_threads(H):
    """Polars may call the plugin from several of its threads at once (is_elementwise=True): thread-local contexts
    and error slots must keep the calls independent."""
    import threading
    A, B = gen.pairs(55, 60000, gen.ASCII_LOWER, 0, 40)
    exp = {m: expect(m, A, B) for m in O.MEASURES}
    errs = []

    def work(m, reps):
        try:
            for _ in range(reps):
                check(H.call_plugin(m, A, B), exp[m])
            with pytest.raises(H.PluginError, match="same length"):
                H.call_plugin(m, A[:3], B[:5])
        except BaseException as e:  # noqa: BLE001
            errs.append((m, repr(e)))

    th = [threading.Thread(target=work, args=(m, 3)) for m in O.MEASURES for _ in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs, errs


@pytest.mark.parametrize("devices,min_rows", [("0,0", "1000"), ("0,0,0", "50000")])
def test_rows_shard_over_several_device_pipelines(H, monkeypatch, devices, min_rows):
    """One plugin call, several device pipelines (the reference fans rows out over its threads inside the call,
    strsim.rs:72-100): the call's slices are dealt out to the pipelines in turn, each with its own context, streams and pinned
    slots, two slices in flight per pipeline.  A one-GPU box runs them as several contexts on device 0; the slice sizes are
    shrunk so that this small frame is a dozen slices; nulls, a literal and strings that need the slow kernels included (their
    kernels are launched when a slice is retired: the column is fetched again); then the same thread goes back to one device."""
    monkeypatch.setenv("POLARS_STRSIM_DEVICES", devices)
    monkeypatch.setenv("POLARS_STRSIM_MIN_ROWS_PER_DEVICE", min_rows)
    monkeypatch.setenv("POLARS_STRSIM_SINGLE_SLICE_ROWS", "10000")
    monkeypatch.setenv("POLARS_STRSIM_RAMP_ROWS", "40000")
    monkeypatch.setenv("POLARS_STRSIM_SLICE_ROWS", "40000")
    monkeypatch.setenv("POLARS_STRSIM_DIRECT_ROWS", "0")
    A, B = gen.pairs(77, 200_003, gen.ASCII_LOWER, 0, 40)
    A2, B2 = gen.pairs(78, 3000, gen.MIXED, 0, 150)
    A, B = A + A2, B + B2
    A[5] = None
    B[100_000] = None
    A[-1] = "z" * 1300
    for m in ("levenshtein", "jaro_winkler", "jaccard"):
        check(H.call_plugin(m, A, B), expect(m, A, B))
    check(H.call_plugin("levenshtein", A, "phillips"), expect("levenshtein", A, ["phillips"]))
    check(H.call_plugin("jaro", "philips", B), expect("jaro", ["philips"], B))
    monkeypatch.setenv("POLARS_STRSIM_DEVICES", "0")
    check(H.call_plugin("sorensen_dice", A[:70000], B[:70000]), expect("sorensen_dice", A[:70000], B[:70000]))
    monkeypatch.delenv("POLARS_STRSIM_DEVICES")  # the default: one device, the calling thread's current one
    check(H.call_plugin("jaccard", A[:70000], B[:70000]), expect("jaccard", A[:70000], B[:70000]))


def test_large_call_with_and_without_pinned_result_and_length_bytes(H, monkeypatch):
    """A multi-slice call (1.3 M rows): the result column in pooled pinned memory (default) or malloc'd + copied, offsets
    shipped as one length byte per row (default) or as u32 -- the four combinations agree bit for bit, and match the oracle on
    windows; a second call reuses the pool's block."""
    A, B = gen.pairs(91, 1_300_000, gen.ASCII_LOWER, 0, 32)
    A[7] = None
    B[1_200_000] = "q" * 300  # one slice of column b falls back to u32 offsets
    ref = None
    for pinned, lens in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("POLARS_STRSIM_PINNED_OUT", pinned)
        monkeypatch.setenv("POLARS_STRSIM_LENGTH_BYTES", lens)
        got = H.call_plugin("levenshtein", A, B)
        if ref is None:
            ref = got
            for lo in (0, 650_000, 1_199_000, 1_290_000):
                check(got[lo:lo + 10_000], expect("levenshtein", A[lo:lo + 10_000], B[lo:lo + 10_000]))
        else:
            assert got.equals(ref), (pinned, lens)


def test_large_call_over_two_pipelines_lands_in_one_pinned_column(H, monkeypatch):
    """4.6 M rows: enough for two device pipelines at the default rows-per-device threshold, and a result column that comes
    from the pinned pool -- both pipelines' copy engines write their slices of it directly.  Equal, bit for bit, to the
    one-pipeline call (whose parity with the oracle the other tests establish)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_support import workload as W
    n = 4_600_000
    _, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
    oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, n)
    mk = lambda o, v: pa.StringArray.from_buffers(n, pa.py_buffer(o.astype(np.int32)), pa.py_buffer(v)).cast(pa.string_view())
    a, b = mk(oa, va), mk(ob, vb)
    for m in ("levenshtein", "jaro_winkler"):
        monkeypatch.setenv("POLARS_STRSIM_DEVICES", "0")
        one = H.call_plugin(m, a, b)
        monkeypatch.setenv("POLARS_STRSIM_DEVICES", "0,0")
        two = H.call_plugin(m, a, b)
        assert two.equals(one), m
    monkeypatch.setenv("POLARS_STRSIM_DEVICES", "0")
    H.call_plugin("levenshtein", a[:1000], b[:1000])  # (this thread's context goes back to one device)


@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("measure", ["levenshtein", "jaro_winkler", "sorensen_dice"])
def test_view_native_slices(H, monkeypatch, measure, parallel):
    """view columns shipped as the views themselves (POLARS_STRSIM_VIEWS=1; the default of the engine-parallel mode)
    and resolved on the device -- inline strings, strings beyond 12 bytes, long ones, non-ASCII, nulls in both columns, chunks at
    odd boundaries, sliced arrays; several slices per call (the estimate of one slice sizes the next), both engine modes."""
    import random
    monkeypatch.setenv("POLARS_STRSIM_VIEWS", "1")
    monkeypatch.setenv("POLARS_STRSIM_DIRECT_ROWS", "0")
    monkeypatch.setenv("POLARS_STRSIM_SINGLE_SLICE_ROWS", "1000")
    monkeypatch.setenv("POLARS_STRSIM_RAMP_ROWS", "16384")
    monkeypatch.setenv("POLARS_STRSIM_SLICE_ROWS", "32768")
    rng = random.Random(19)
    A, B = gen.pairs(41, 60_000, gen.ASCII_LOWER, 0, 11)      # every string in its view
    A2, B2 = gen.pairs(42, 30_000, gen.ASCII_LOWER, 0, 40)    # in and out of the views
    A3, B3 = gen.pairs(43, 3_000, gen.MIXED, 0, 60)
    A4, B4 = gen.pairs(44, 300, gen.ASCII_LOWER, 100, 1500)
    A, B = A + A2 + A3 + A4, B + B2 + B3 + B4
    order = list(range(len(A)))
    rng.shuffle(order)
    A, B = [A[i] for i in order], [B[i] for i in order]
    A = [None if rng.random() < 0.05 else x for x in A]
    B = [None if rng.random() < 0.05 else x for x in B]
    pa_a, pa_b = pa.array(A, type=pa.string()), pa.array(B, type=pa.string())
    ca = pa.chunked_array([pa_a[:7], pa_a[7:20_001], pa_a[20_001:20_001], pa_a[20_001:]])
    cb = pa.chunked_array([pa_b[:50_000], pa_b[50_000:]])
    exp = expect(measure, A, B)
    check(H.call_plugin(measure, ca, cb, layout="vu", parallel=parallel), exp)
    check(H.call_plugin(measure, pa_a[1234:80_000], pa_b[1234:80_000], layout="vu", parallel=parallel), exp[1234:80_000])
    # one view column, one offsets column: the call packs as before
    check(H.call_plugin(measure, ca, cb, layout=("vu", "u"), parallel=parallel), exp)
