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
    ctx = S.Context(0, stream=torch.cuda.current_stream().cuda_stream)
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
