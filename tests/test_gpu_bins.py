"""GPU parity of the BINNED path for rows of 33..128 ASCII bytes (csrc/strsim_bins.h, strsim_lane_bins.h): k_lane_stage hands
those rows to k_wide_bins through pages of 64 records of one bin.  A context takes that path when its previous call left many such
rows, so every test runs a frame twice on one context -- first through the mask-driven kernels, then binned -- and holds both
to the oracle bit for bit (reference semantics: strsim.rs:141-160, :200-237, :257-270, :297-305, :333-341).
"""
import os
import random

import numpy as np
import pytest

import gen
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import strsim_amd
    return strsim_amd


@pytest.fixture(autouse=True)
def small_frames_are_binned(monkeypatch):
    monkeypatch.setenv("STRSIM_BINS_MIN_ROWS", "1")  # (read per call by the library; default: frames of 2^20 rows and more)


def u64(x):
    return np.asarray(x, dtype=np.float64).view(np.uint64)


def assert_bit_exact(got, exp, A, B, what):
    bad = np.nonzero(u64(got) != u64(exp))[0]
    if bad.size:
        i = int(bad[0])
        raise AssertionError(f"{what}: {bad.size}/{len(exp)} rows differ; first row {i}: a={A[i]!r} b={B[i]!r} "
                             f"got={got[i]!r} exp={exp[i]!r}")


def to_device(S, A, B):
    import torch
    oa, va = S.pack_strings(A)
    ob, vb = S.pack_strings(B)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    cols = (t(oa, np.int32), t(np.concatenate([va, pad]), np.uint8), t(ob, np.int32), t(np.concatenate([vb, pad]), np.uint8))
    torch.cuda.synchronize()
    return cols


def candidate(a, b):
    la, lb = len(a.encode()), len(b.encode())
    return 32 < max(la, lb) <= 128 and min(la, lb) >= 1


def mixed_frame(seed, n):
    """Every kind of row the binned kernel must take or leave: lengths 1..128 on both sides (independent, edited and equal
    pairs), mixed case (seven planes), short against long, the corner lengths of every bin edge, and -- to be left alone --
    non-ASCII rows, rows beyond 128 bytes and empty strings; not a multiple of 64 rows."""
    rng = random.Random(seed)
    A, B = gen.pairs(seed, n, gen.ASCII_LOWER, 1, 128)
    A2, B2 = gen.pairs(seed + 1, n // 8, gen.ASCII_LOWER + "ABCDEFGH _-", 20, 128)
    A3, B3 = gen.pairs(seed + 2, n // 16, gen.MIXED, 0, 60)
    A4, B4 = gen.pairs(seed + 3, n // 64, gen.ASCII_LOWER, 100, 300)
    edge = [1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 79, 80, 81, 95, 96, 97, 111, 112, 113, 127, 128]
    A5, B5 = [], []
    for la in edge:
        for lb in edge:
            A5.append(gen.rand_string(rng, "abc", la, la))
            B5.append(gen.rand_string(rng, "abc", lb, lb))
    A6 = ["", "x" * 40, "", "é" * 30 + "a" * 20, "a" * 128, "ab" * 64]
    B6 = ["y" * 50, "", "", "a" * 50, "a" * 128, "ba" * 64]
    rows = list(zip(A + A2 + A3 + A4 + A5 + A6, B + B2 + B3 + B4 + B5 + B6))
    rng.shuffle(rows)
    if len(rows) % 64 == 0:
        rows.append(("k" * 70, "k" * 69))
    return [r[0] for r in rows], [r[1] for r in rows]


@pytest.mark.parametrize("measure", O.MEASURES)
def test_binned_rows_match_the_oracle(S, measure):
    A, B = mixed_frame(4100 + O.MEASURES.index(measure), 12_000)
    exp = O.batch_strings(measure, A, B, 8)
    cols = to_device(S, A, B)
    ncand_ascii = sum(1 for a, b in zip(A, B) if candidate(a, b) and a.isascii() and b.isascii())
    with S.Context(0) as ctx:
        out = ctx.pairs_device(measure, *cols)
        ctx.synchronize()
        assert ctx.last_binned_rows == 0  # a fresh context knows nothing about the frame: the mask-driven kernels
        assert_bit_exact(out.cpu().numpy(), exp, A, B, measure + ", first call")
        binned = []
        for rep in range(3):
            out.fill_(-1.0)
            out = ctx.pairs_device(measure, *cols, out=out)
            ctx.synchronize()
            binned.append(ctx.last_binned_rows)
            assert_bit_exact(out.cpu().numpy(), exp, A, B, "%s, call %d (%d rows binned)" % (measure, rep + 1, binned[-1]))
        # every candidate goes through the bins (k_wide_bins hands the non-ASCII ones back through the mask)
        assert binned[0] == binned[1] == binned[2] >= ncand_ascii, (binned, ncand_ascii)


@pytest.mark.parametrize("measure", ["levenshtein", "jaro_winkler"])
def test_frames_of_every_density_and_size(S, measure):
    """Groups of 2 048 rows with no candidate at all, with nothing but candidates, a frame smaller than one group, one that
    ends inside a group; candidates in a few bins only."""
    short = gen.pairs(51, 5000, gen.ASCII_LOWER, 1, 30, max_bytes=30)
    long_ = gen.pairs(52, 5000, gen.ASCII_LOWER, 90, 128, max_bytes=128)
    mid = gen.pairs(53, 3000, gen.ASCII_LOWER, 40, 44, max_bytes=44)
    frames = {
        "blocks of short and long": (short[0][:2500] + long_[0] + short[0][2500:] + mid[0], short[1][:2500] + long_[1] + short[1][2500:] + mid[1]),
        "smaller than a group": (long_[0][:700] + short[0][:300], long_[1][:700] + short[1][:300]),
        "all candidates": (long_[0][:4097], long_[1][:4097]),
    }
    for what, (A, B) in frames.items():
        exp = O.batch_strings(measure, A, B, 8)
        cols = to_device(S, A, B)
        with S.Context(0) as ctx:
            binned = []
            for rep in range(4):
                out = ctx.pairs_device(measure, *cols)
                ctx.synchronize()
                binned.append(ctx.last_binned_rows)
                assert_bit_exact(out.cpu().numpy(), exp, A, B, "%s, %s, call %d" % (measure, what, rep))
            assert binned[0] == 0 and binned[1] > 0 and binned[3] == binned[2] == binned[1], (what, binned)


def test_the_longest_pairs_and_the_last_rows_of_the_columns(S):
    """128-byte pairs (the widest masks, eight pieces a side); the frame's last rows end at the columns' last bytes, where a
    row's last 16-byte piece reaches past the column (read byte by byte there)."""
    A, B = gen.pairs(61, 3000, gen.ASCII_LOWER, 120, 128, max_bytes=128)
    A += ["q" * 33, "r" * 37]
    B += ["q" * 34, "rs" * 17]
    import torch
    oa, va = S.pack_strings(A)
    ob, vb = S.pack_strings(B)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    cols = (t(oa, np.int32), t(va, np.uint8), t(ob, np.int32), t(vb, np.uint8))  # (no padding behind the columns)
    torch.cuda.synchronize()
    for measure in ("jaccard", "jaro"):
        exp = O.batch_strings(measure, A, B, 8)
        with S.Context(0) as ctx:
            seen = []
            for rep in range(3):
                out = ctx.pairs_device(measure, *cols)
                ctx.synchronize()
                seen.append(ctx.last_binned_rows)
                assert_bit_exact(out.cpu().numpy(), exp, A, B, "%s, call %d" % (measure, rep))
            assert seen == [0, len(A), len(A)], seen


def test_the_context_falls_back_when_the_long_rows_are_gone(S):
    """long rows -> binned; then a frame of short rows only (binned mode finds nothing to bin: everything computed in place) ->
    the context goes back to the plain kernel."""
    L = gen.pairs(71, 4000, gen.ASCII_LOWER, 33, 100, max_bytes=100)
    Sh = gen.pairs(72, 4000, gen.ASCII_LOWER, 1, 32, max_bytes=32)
    cl, cs = to_device(S, *L), to_device(S, *Sh)
    el, es = O.batch_strings("sorensen_dice", L[0], L[1], 8), O.batch_strings("sorensen_dice", Sh[0], Sh[1], 8)
    with S.Context(0, one_launch=True) as ctx:
        for what, cols, exp, AB in (("long", cl, el, L), ("long", cl, el, L), ("short", cs, es, Sh), ("short", cs, es, Sh), ("short", cs, es, Sh)):
            before = ctx.enqueued_ops
            out = ctx.pairs_device("sorensen_dice", *cols)
            ops = ctx.enqueued_ops - before
            ctx.synchronize()
            assert_bit_exact(out.cpu().numpy(), exp, AB[0], AB[1], what)
        assert ops == 1 and ctx.last_binned_rows == 0  # back to one launch per call
