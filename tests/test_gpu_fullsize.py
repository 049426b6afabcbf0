"""Parity at BASELINE.json's full size (cfg2: 100 M rows, <= 32-byte strings) through size-independent
properties, plus oracle spot checks on row windows regenerated on the CPU (the synthetic frame is counter-based,
so any window can be rebuilt independently -- bench_support/synth.h)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROWS = int(os.environ.get("STRSIM_FULLSIZE_ROWS", "100000000"))


@pytest.fixture(scope="module")
def frame():
    import torch
    import strsim_amd as S
    from bench_support import workload as W
    _, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
    dev = torch.device("cuda", 0)
    offA, valA, offB, valB, ba, bb = W.device_columns(seed, law, lo, hi, 0, ROWS, dev)
    torch.cuda.synchronize()  # the generator's kernels ran on torch's stream; the context owns its stream (run() waits for it)
    ctx = S.Context(0)
    yield dict(W=W, S=S, ctx=ctx, cols=(offA, valA, offB, valB), cfg=(seed, law, lo, hi), torch=torch)
    ctx.close()


def run(frame, measure, swap=False):
    offA, valA, offB, valB = frame["cols"]
    ctx = frame["ctx"]
    out = ctx.pairs_device(measure, offB, valB, offA, valA) if swap else ctx.pairs_device(measure, offA, valA, offB, valB)
    ctx.synchronize()
    frame["torch"].cuda.synchronize()
    return out


@pytest.mark.parametrize("measure", ["levenshtein", "jaccard", "sorensen_dice"])
def test_symmetric_measures_are_bitwise_symmetric(frame, measure):
    t = frame["torch"]
    x = run(frame, measure)
    y = run(frame, measure, swap=True)
    assert t.equal(x.view(t.int64), y.view(t.int64))
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    assert not bool(t.isnan(x).any())


@pytest.mark.parametrize("measure", O.MEASURES)
def test_identity_and_bounds(frame, measure):
    t = frame["torch"]
    offA, valA, offB, valB = frame["cols"]
    ctx = frame["ctx"]
    same = ctx.pairs_device(measure, offA, valA, offA, valA)
    ctx.synchronize()
    t.cuda.synchronize()
    assert bool((same == 1.0).all())  # a == b -> 1.0 (strsim.rs:128,182,288,324)
    x = run(frame, measure)
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    la = (offA[1:] - offA[:-1])
    lb = (offB[1:] - offB[:-1])
    if measure == "levenshtein":
        # distance >= |la - lb|  <=>  similarity <= 1 - |la-lb|/max(la,lb)
        mx = t.maximum(la, lb).to(t.float64)
        ub = 1.0 - (la - lb).abs().to(t.float64) / mx
        assert bool((x <= ub + 1e-15).all())
    # a checksum that is deterministic across runs
    y = run(frame, measure)
    assert t.equal(x.view(t.int64), y.view(t.int64))


@pytest.mark.parametrize("measure", O.MEASURES)
def test_windows_match_the_oracle(frame, measure):
    W = frame["W"]
    seed, law, lo, hi = frame["cfg"]
    x = run(frame, measure)
    rng = np.random.default_rng(5)
    starts = [0, ROWS - 50000] + [int(s) for s in rng.integers(0, ROWS - 50000, 6)]
    for s in starts:
        oa, va, ob, vb = W.host_columns(seed, law, lo, hi, s, 50000)
        exp = O.batch(measure, oa, va, ob, vb, nthreads=8)
        got = x[s:s + 50000].cpu().numpy()
        bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
        assert bad.size == 0, (measure, s, int(bad[0]), got[bad[0]], exp[bad[0]])


def _cores():
    n = os.cpu_count() or 1
    try:  # (the GPU boxes: a cgroup quota of 16 CPUs on 256 logical ones -- more threads than that only get throttled)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(round(int(q) / int(per)))))
    except Exception:
        pass
    return min(n, 32)


def _every_row_against_the_oracle(measure, parts, out, chunk=8_000_000):
    """EVERY row of a frame against the oracle (full-size parity was by windows).  The oracle reads the very bytes
    the kernels read -- the device columns copied to the host, `chunk` rows at a time -- so generator and transfer cannot hide a
    difference; that the device generator equals the host one is what the window tests (host-generated) hold."""
    cores, bad_total, rows_total = _cores(), 0, 0
    for r0, r1, oa, va, ob, vb in parts:
        n = r1 - r0
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            ha = oa[c0:c1 + 1].cpu().numpy().view(np.uint32)
            hb = ob[c0:c1 + 1].cpu().numpy().view(np.uint32)
            hva = va[int(ha[0]):int(ha[-1])].cpu().numpy()
            hvb = vb[int(hb[0]):int(hb[-1])].cpu().numpy()
            exp = O.batch(measure, ha - ha[0], hva, hb - hb[0], hvb, nthreads=cores)
            got = out[r0 + c0:r0 + c1].cpu().numpy()
            bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
            assert bad.size == 0, (measure, r0 + c0 + int(bad[0]), got[bad[0]], exp[bad[0]], int(bad.size))
            rows_total += c1 - c0
    return rows_total


@pytest.mark.parametrize("measure", O.MEASURES)
def test_every_row_of_cfg2_matches_the_oracle(frame, measure):
    """cfg2 at full size, all 100 M rows, bit for bit, for each of the five measures."""
    offA, valA, offB, valB = frame["cols"]
    x = run(frame, measure)
    assert _every_row_against_the_oracle(measure, [(0, ROWS, offA, valA, offB, valB)], x) == ROWS


# ---------------------------------------------------------------------------------------------------------------
# The other BASELINE.json configs at their full size, on the frames bench.py times (bench_support/workload.py):
# cfg3 (Jaro-Winkler, 100 M rows, Zipf 4..128 bytes), cfg5 (Levenshtein, 10 M rows, U{1..1024} bytes, held as two row
# batches because one column is ~5 GB > 32-bit offsets) and cfg4 (all five measures, 200 M rows, the fused call) on one
# GPU.  Size-independent properties on every row + oracle windows regenerated on the CPU.
# ---------------------------------------------------------------------------------------------------------------
SCALE = float(os.environ.get("STRSIM_FULLSIZE_SCALE", "1.0"))  # < 1: smaller frames for a quick local run


def _device_frame(name, rows):
    import torch
    from bench_support import workload as W
    measure, _, law, lo, hi, seed = W.CONFIGS[name]
    mean_len = (lo + hi) / 2.0 if law == W.UNIFORM else 26.0
    nparts = max(1, int(rows * mean_len * 1.15 / 3.5e9) + (1 if rows * mean_len * 1.15 > 3.5e9 else 0))
    bounds = [rows * p // nparts for p in range(nparts + 1)]
    dev = torch.device("cuda", 0)
    parts = []
    for p in range(nparts):
        r0, r1 = bounds[p], bounds[p + 1]
        oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, r0, r1 - r0, dev)
        parts.append((r0, r1, oa, va, ob, vb))
    return measure, (seed, law, lo, hi), parts


def _oracle_windows(measure, cfg, rows, out, win, nwin, seed=11):
    from bench_support import workload as W
    s_, law, lo, hi = cfg
    rng = np.random.default_rng(seed)
    starts = [0, rows - win] + [int(s) for s in rng.integers(0, rows - win, nwin)]
    for s in starts:
        oa, va, ob, vb = W.host_columns(s_, law, lo, hi, s, win)
        exp = O.batch(measure, oa, va, ob, vb, nthreads=8)
        got = out[s:s + win].cpu().numpy()
        bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
        assert bad.size == 0, (measure, s, int(bad[0]), got[bad[0]], exp[bad[0]])


@pytest.mark.parametrize("name,rows,win,nwin", [("cfg3", 100_000_000, 20000, 6), ("cfg5", 10_000_000, 1500, 3)])
def test_config_frame_full_size(name, rows, win, nwin):
    import torch as t
    import strsim_amd as S
    rows = max(int(rows * SCALE), 4 * win)
    measure, cfg, parts = _device_frame(name, rows)
    t.cuda.synchronize()  # the frame is complete; the context's own stream is waited for inside run()
    with S.Context(0) as ctx:
        def run(swap=False, same=False):
            out = t.empty(rows, dtype=t.float64, device="cuda")
            for r0, r1, oa, va, ob, vb in parts:
                if same:
                    ctx.pairs_device(measure, oa, va, oa, va, out=out[r0:r1])
                elif swap:
                    ctx.pairs_device(measure, ob, vb, oa, va, out=out[r0:r1])
                else:
                    ctx.pairs_device(measure, oa, va, ob, vb, out=out[r0:r1])
            ctx.synchronize()
            t.cuda.synchronize()
            return out
        x = run()
        assert not bool(t.isnan(x).any()) and float(x.min()) >= 0.0 and float(x.max()) <= 1.0
        assert t.equal(x.view(t.int64), run().view(t.int64))                 # deterministic, bit for bit
        assert bool((run(same=True) == 1.0).all())                            # a == b -> 1.0 (strsim.rs:128, :182)
        if measure == "levenshtein":
            assert t.equal(x.view(t.int64), run(swap=True).view(t.int64))     # distance is symmetric
        _oracle_windows(measure, cfg, rows, x, win, nwin)
        if name == "cfg3":  # (cfg5's 2.6e12 DP cells are hours of oracle time: windows there)
            assert _every_row_against_the_oracle(measure, parts, x) == rows


def test_cfg4_fused_five_outputs_full_size():
    """cfg4 on one GPU: 200 M rows, one fused pass with five outputs (strsim_pairs_device_all) -- every output column equals,
    bit for bit, the single-measure call on the same frame, and oracle windows match."""
    import torch as t
    import strsim_amd as S
    rows = max(int(200_000_000 * SCALE), 200_000)
    _, cfg, parts = _device_frame("cfg4", rows)
    t.cuda.synchronize()
    with S.Context(0) as ctx:
        outs = [t.empty(rows, dtype=t.float64, device="cuda") for _ in S.MEASURES]
        for r0, r1, oa, va, ob, vb in parts:
            ctx.pairs_device_all(oa, va, ob, vb, outs=[o[r0:r1] for o in outs])
        ctx.synchronize()
        t.cuda.synchronize()
        single = t.empty(rows, dtype=t.float64, device="cuda")
        for k, m in enumerate(S.MEASURES):
            for r0, r1, oa, va, ob, vb in parts:
                ctx.pairs_device(m, oa, va, ob, vb, out=single[r0:r1])
            ctx.synchronize()
            t.cuda.synchronize()
            assert t.equal(outs[k].view(t.int64), single.view(t.int64)), m
            _oracle_windows(m, cfg, rows, outs[k], 20000, 2, seed=40 + k)
