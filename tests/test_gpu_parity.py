"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Bar: bit-exact f64 for all five measures (integer intermediates are exact; the epilogues use the
reference's IEEE operation order with contraction off).  north_star allows 1 ulp for the Jaro family;
we hold it to 0.
"""
import struct

import numpy as np
import pytest

import gen
import oracle_lib as O
from golden_data import reference_vectors, readme_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import strsim_amd
    return strsim_amd


@pytest.fixture(scope="module")
def ctx(S):
    c = S.Context(0)
    yield c
    c.close()


def u64(x):
    return np.asarray(x, dtype=np.float64).view(np.uint64)


def assert_bit_exact(got, exp, A, B, what):
    bad = np.nonzero(u64(got) != u64(exp))[0]
    if bad.size:
        i = int(bad[0])
        raise AssertionError(f"{what}: {bad.size}/{len(exp)} rows differ; first row {i}: "
                             f"a={A[i if len(A) > 1 else 0]!r} b={B[i if len(B) > 1 else 0]!r} got={got[i]!r} exp={exp[i]!r}")


def gpu(S, ctx, m, A, B):
    ao, av = S.pack_strings(A)
    bo, bv = S.pack_strings(B)
    return ctx.pairs_host(m, ao, av, bo, bv)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_reference_vectors(S, ctx, measure):
    rows = [r for r in reference_vectors() if r[0] == measure]
    A = [r[2] for r in rows]
    B = [r[3] for r in rows]
    exp = np.array([r[4] for r in rows])
    got = gpu(S, ctx, measure, A, B)
    assert np.all(np.abs(got - exp) < 1e-8)  # strsim.rs:350
    assert_bit_exact(got, O.batch_strings(measure, A, B), A, B, measure)


def test_readme_table(S):
    rows = readme_table()
    A = [r["name_a"] for r in rows]
    B = [r["name_b"] for r in rows]
    for m in O.MEASURES:
        got = getattr(S, m)(A, B)
        for r, g in zip(rows, got):
            if r[m] is None:
                assert np.isnan(g)
            else:
                assert abs(g - r[m]) < 5e-7


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("alphabet,lo,hi", [("ab", 0, 32), (gen.ASCII_LOWER, 0, 32), (gen.ASCII_LOWER, 1, 12),
                                             ("".join(chr(c) for c in range(1, 128)), 0, 32)])
def test_lane_path_random(S, ctx, measure, alphabet, lo, hi):
    A, B = gen.pairs(hash((measure, alphabet, hi)) & 0xFFFF, 20000, alphabet, lo, hi, max_bytes=32)
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)
    assert ctx.last_wave_rows <= 64  # only rows whose 32-byte window crosses the end of the buffer


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("alphabet", ["абвгдежзийклмнопрстуфхцчшщыэюя ", "abcdeéèüñöçß-'", "日本語中文字漢한국어テキスト", "αβγ abc жзи 語"])
def test_short_non_ascii_lane_path(S, ctx, measure, alphabet):
    """<= 32 scalar values per string in any BMP script: decoded per lane (k_lane_utf8), not one wave per pair."""
    A, B = gen.pairs(hash((measure, alphabet)) & 0xFFFF, 12000, alphabet, 0, 30)
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)
    leftovers = sum(1 for a, b in zip(A, B) if not a or not b or len(a) > 32 or len(b) > 32
                    or len(a.encode()) > 128 or len(b.encode()) > 128)
    assert ctx.last_wave_rows <= leftovers + 64


@pytest.mark.parametrize("measure", ["levenshtein", "jaro_winkler"])
def test_non_ascii_eligibility_edges(S, ctx, measure):
    """exactly 32 / 33 scalar values, 4-byte (astral) values, 128 / 129 bytes: the lane path must hand over cleanly."""
    cases = [("я" * 32, "я" * 31 + "ю"), ("я" * 33, "я" * 33), ("я" * 33, "ю"), ("語" * 32, "語" * 30 + "漢字"),
             ("語" * 42, "語" * 43), ("語" * 43, "語" * 42 + "x"), ("a😀b", "a😀c"), ("😀" * 8, "😀" * 7 + "𝄞"),
             ("é" * 64, "é" * 64), ("é" * 64 + "e", "é" * 64), ("naïve", "naive"), ("Ünïcödé", "Unicode"), ("ß", "ss")]
    A = [c[0] for c in cases] * 40
    B = [c[1] for c in cases] * 40
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 4), A, B, measure)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_embedded_nul_and_control_bytes(S, ctx, measure):
    """A Rust &str may hold any scalar value, U+0000 included."""
    A, B = gen.pairs(91, 8000, "ab\x00\x01\x7f ", 0, 32)
    A2, B2 = gen.pairs(92, 1000, "ab\x00\x7fé", 0, 80)
    A, B = A + A2, B + B2
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_wave_path_unicode_and_long(S, ctx, measure):
    A, B = gen.pairs(11, 3000, gen.MIXED, 0, 40)
    A2, B2 = gen.pairs(12, 600, gen.ASCII_LOWER, 20, 300)
    A3, B3 = gen.pairs(13, 200, gen.MIXED, 100, 250, max_bytes=1024)
    A, B = A + A2 + A3, B + B2 + B3
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)
    assert ctx.last_wave_rows > 1000


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("lo,hi,alphabet", [(33, 64, gen.ASCII_LOWER), (65, 128, gen.ASCII_LOWER), (1, 128, "ab"),
                                            (20, 128, "".join(chr(c) for c in range(1, 128)))])
def test_wide_lane_path_random(S, ctx, measure, lo, hi, alphabet):
    """33..128-byte ASCII strings: the W = 2 / W = 4 lane kernel (k_lane_wide)."""
    A, B = gen.pairs(hash((measure, lo, hi)) & 0xFFFF, 6000, alphabet, lo, hi, max_bytes=128)
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)
    assert ctx.last_wave_rows <= 64 + sum(1 for a, b in zip(A, B) if not a or not b)


@pytest.mark.parametrize("alphabet", [gen.ASCII_LOWER, "ab", "".join(chr(c) for c in range(1, 128))])
def test_long_ascii_levenshtein_blocks(S, ctx, alphabet):
    """129..1024-byte ASCII strings: block-parallel Myers across lanes (wave_lev_blocks)."""
    A, B = gen.pairs(len(alphabet), 1500, alphabet, 1, 1024, max_bytes=1024)
    A2, B2 = gen.pairs(len(alphabet) + 1, 500, alphabet, 900, 1024, max_bytes=1024)
    A, B = A + A2, B + B2
    got = gpu(S, ctx, "levenshtein", A, B)
    assert_bit_exact(got, O.batch_strings("levenshtein", A, B, 8), A, B, "levenshtein")


def test_long_ascii_levenshtein_many_batches_per_wave(S, monkeypatch):
    """Few resident waves (one per CU), so every wave runs many batches and takes several chunks from the work list:
    the staged-text slots (global scratch) and the LDS match table are reused over and over."""
    monkeypatch.setenv("STRSIM_LEV_WAVES_PER_CU", "1")
    c = S.Context(0)
    try:
        A, B = gen.pairs(77, 24000, gen.ASCII_LOWER, 100, 1024, max_bytes=1024)
        A2, B2 = gen.pairs(78, 6000, gen.ASCII_LOWER + gen.ASCII_LOWER.upper() + " .,", 100, 1024, max_bytes=1024)
        A, B = A + A2, B + B2
        got = gpu(S, c, "levenshtein", A, B)
        assert_bit_exact(got, O.batch_strings("levenshtein", A, B, 16), A, B, "levenshtein")
    finally:
        c.close()


@pytest.mark.parametrize("waves", ["1", "20"])
def test_long_levenshtein_rows_pooled_across_chunks(S, monkeypatch, waves):
    """k_wave_pairs<levenshtein> ranks the rows of several 64-row chunks together before it deals them into batches: long rows
    that are sparse (a few per chunk, so a pool spans many chunks), dense stretches (a pool is four full chunks), and every kind
    of row in the same pool -- ASCII with 64-row blocks, long Cyrillic / CJK (32-row blocks of symbols), an empty side, strings
    past 1 024 bytes (left to the long-string pass) -- against the oracle."""
    import random
    monkeypatch.setenv("STRSIM_LEV_WAVES_PER_CU", waves)
    rng = random.Random(4242)
    def text(alpha, lo, hi):
        return "".join(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))
    A, B = [], []
    for i in range(30000):
        dense = 9000 <= i < 11000
        r = rng.random()
        if dense or r < 0.03:
            kind = rng.random()
            if kind < 0.70:
                A.append(text("abcdefghijklmnopqrstuvwxyz", 1, 1024)); B.append(text("abcdefghijklmnopqrstuvwxyz", 129, 1024))
            elif kind < 0.80:
                A.append(text("abcXYZ 019.,", 129, 700)); B.append(text("abcXYZ 019.,", 1, 700))
            elif kind < 0.88:
                A.append(text("абвгдежзийклмн", 70, 400)); B.append(text("абвгдежзийклмн", 70, 400))
            elif kind < 0.92:
                A.append(text("日本語の文字列漢字", 50, 300)); B.append(text("abc日本", 50, 300))
            elif kind < 0.96:
                A.append(""); B.append(text("abc", 129, 900))
            else:
                A.append(text("ab", 1025, 1500)); B.append(text("ab", 200, 1100))
        else:
            A.append(text("abcdefgh", 0, 30)); B.append(text("abcdefgh", 0, 30))
    c = S.Context(0)
    try:
        got = gpu(S, c, "levenshtein", A, B)
        assert_bit_exact(got, O.batch_strings("levenshtein", A, B, 16), A, B, "levenshtein")
    finally:
        c.close()


@pytest.mark.parametrize("density", [0.3, 0.55, 1.0])
@pytest.mark.parametrize("measure", O.MEASURES)
def test_wide_rows_at_every_list_density(S, ctx, measure, density):
    """k_lane_wide sorts the 33..128-byte rows of a 16 384-row super-span into one list when they fit it (8 192 rows), else half a
    super at a time: frames with 30 % of such rows take the first path, with 55 % and 100 % the second (after a counting pass that
    found them too many)."""
    import random
    rng = random.Random(int(density * 100))
    A, B = [], []
    for _ in range(40000):
        if rng.random() < density:
            la, lb = rng.randint(1, 128), rng.randint(33, 128)
            if rng.random() < 0.5:
                la, lb = lb, la
        else:
            la, lb = rng.randint(0, 32), rng.randint(0, 32)
        a = "".join(rng.choice("abcdefghij") for _ in range(la))
        # half of the pairs are edited copies (matches, transpositions, common prefixes for Jaro-Winkler)
        if rng.random() < 0.5 and la and lb:
            b = list(a[:lb]) + [rng.choice("abcdefghij") for _ in range(max(0, lb - la))]
            for _ in range(rng.randint(0, 4)):
                b[rng.randrange(len(b))] = rng.choice("abcxyz")
            b = "".join(b)
        else:
            b = "".join(rng.choice("abcdefghij") for _ in range(lb))
        A.append(a); B.append(b)
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 16), A, B, measure)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_wide_rows_mask_and_text_width_classes(S, ctx, measure):
    """k_lane_wide sizes its masks by the PATTERN (b for Jaro / Jaro-Winkler, the longer string for the symmetric measures) and
    fetches the text by its own length: every (text class, pattern class) pair of 32-byte classes, in a frame large enough for
    rounds whose windows the wave fetches together (the last rows of the columns take the lane-by-lane path), with copies and
    transposed copies so that Jaro finds matches at every distance (strsim.rs:200-237)."""
    import random
    rng = random.Random(77)
    edges = (1, 2, 16, 31, 32, 33, 48, 63, 64, 65, 80, 95, 96, 97, 112, 127, 128)
    A, B = [], []
    for rep in range(24):
        for la in edges:
            for lb in edges:
                if max(la, lb) <= 32 and rep:  # (the 32-byte kernel's rows: once is enough here)
                    continue
                a = [rng.choice("abcdefghijklmnop") for _ in range(la)]
                kind = rng.random()
                if kind < 0.35:
                    b = [rng.choice("abcdefghijklmnop") for _ in range(lb)]
                elif kind < 0.7:  # an edited copy
                    b = (a * (lb // max(la, 1) + 1))[:lb]
                    for _ in range(rng.randint(0, 5)):
                        b[rng.randrange(lb)] = rng.choice("abcxyz")
                else:  # a copy with neighbours swapped: transpositions
                    b = (a * (lb // max(la, 1) + 1))[:lb]
                    for k in range(0, lb - 1, rng.choice((2, 3, 5))):
                        b[k], b[k + 1] = b[k + 1], b[k]
                A.append("".join(a)); B.append("".join(b))
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 16), A, B, measure)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_length_class_boundaries(S, ctx, measure):
    import random
    rng = random.Random(5)
    A, B = [], []
    for la in (0, 1, 2, 3, 12, 13, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1023, 1024):
        for lb in (0, 1, 2, 31, 32, 33, 64, 65, 128, 1024):
            A.append("".join(rng.choice("abc") for _ in range(la)))
            B.append("".join(rng.choice("abc") for _ in range(lb)))
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 1000])
def test_row_counts(S, ctx, measure, n):
    A, B = gen.pairs(n, n, gen.ASCII_LOWER, 0, 20)
    got = gpu(S, ctx, measure, A, B)
    assert got.shape == (n,)
    assert_bit_exact(got, O.batch_strings(measure, A, B), A, B, measure)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_literal_broadcast(S, ctx, measure):
    A, B = gen.pairs(3, 5000, gen.ASCII_LOWER, 0, 32)
    # literals of every class: short, empty, one byte, exactly 32 bytes, odd / even lengths, mixed case (seven bit-planes),
    # non-ASCII and longer than the lane path takes (both leave the rows to the later kernels)
    for lit in ("phillips", "", "q", "abcdefghijklmnopqrstuvwxyzabcdef", "philips", "McDonald's", "é", "x" * 40):
        got = gpu(S, ctx, measure, A, [lit])
        assert_bit_exact(got, O.batch_strings(measure, A, [lit], 4), A, [lit], measure + " col,lit")
        got = gpu(S, ctx, measure, [lit], B)
        assert_bit_exact(got, O.batch_strings(measure, [lit], B, 4), [lit], B, measure + " lit,col")


@pytest.mark.parametrize("measure", O.MEASURES)
def test_literal_against_every_row_class(S, ctx, measure):
    """The column-x-literal kernel (k_lane_lit<M>; for Jaro / Jaro-Winkler only a literal a goes there, a literal b runs in
    k_lane_stage) on a frame that mixes everything: short ASCII, empty strings, non-ASCII, 33..900-byte rows (blocks get cut to
    the staging area), more rows than one range; literal on either side; empty, long and non-ASCII literals."""
    A, _ = gen.pairs(61, 150_000, gen.ASCII_LOWER, 0, 32)
    A2, _ = gen.pairs(62, 3000, gen.MIXED, 0, 120)
    A3, _ = gen.pairs(63, 300, gen.ASCII_LOWER, 200, 900)
    A = A[:70_000] + A2 + A[70_000:] + A3 + ["", "phillips", "a" * 32]
    for lit in ("phillips", "sm", "abcdefghijklmnopqrstuvwxyzabcde", "", "a", "Phillips-Van Heusen 1881", "m\u00fcller", "x" * 40):
        assert_bit_exact(gpu(S, ctx, measure, A, [lit]), O.batch_strings(measure, A, [lit], 8), A, [lit], measure + " col,lit " + lit)
        assert_bit_exact(gpu(S, ctx, measure, [lit], A), O.batch_strings(measure, [lit], A, 8), [lit], A, measure + " lit,col " + lit)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_literal_of_a_whole_window_against_longer_rows(S, ctx, measure):
    """A literal of exactly 32 / 64 / 96 / 128 bytes is a column of its own whose windows end where the column ends; against rows of
    33..128 bytes k_lane_wide takes its text or its pattern from it -- per LANE for the symmetric measures, and the lanes of a
    round without a row read their columns' first bytes.  (Round 4: the test "does a window reach past its column" was taken per
    lane, the wave's fetch ran with some lanes switched off and handed out null addresses: found by tests/fuzz_gpu.py.)"""
    import random
    rng = random.Random(32)
    for n, lo, hi in ((7000, 0, 40), (3000, 20, 140), (40, 33, 40)):
        A = ["".join(rng.choice("ab") for _ in range(rng.randint(lo, hi))) for _ in range(n)]
        for ll in (31, 32, 33, 64, 96, 128):
            lit = "".join(rng.choice("ab") for _ in range(ll))
            assert_bit_exact(gpu(S, ctx, measure, A, [lit]), O.batch_strings(measure, A, [lit], 8), A, [lit], f"{measure} col,lit{ll}")
            assert_bit_exact(gpu(S, ctx, measure, [lit], A), O.batch_strings(measure, [lit], A, 8), [lit], A, f"{measure} lit{ll},col")


@pytest.mark.parametrize("measure", O.MEASURES)
def test_strings_beyond_the_wave_kernel_cap(S, ctx, measure):
    """> 1024-byte strings: finished by the second pass launched from strsim_ctx_synchronize()."""
    import random
    rng = random.Random(41)
    A, B = gen.pairs(31, 300, gen.ASCII_LOWER, 0, 40)
    specs = [(1025, 1025, gen.ASCII_LOWER), (2048, 1500, "ab"), (2049, 10, gen.ASCII_LOWER), (3000, 2800, gen.ASCII_LOWER),
             (5, 4000, gen.ASCII_LOWER), (1200, 1100, gen.MIXED), (0, 1500, gen.ASCII_LOWER), (1500, 1500, "a")]
    for la, lb, alpha in specs:
        a = "".join(rng.choice(alpha) for _ in range(la))
        b = gen.edit(rng, a, alpha, 3) if la == lb and la > 0 else "".join(rng.choice(alpha) for _ in range(lb))
        pos = rng.randrange(len(A))
        A.insert(pos, a)
        B.insert(pos, b)
    A.append(A[10] * 60)  # a == b, long
    B.append(A[-1])
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)
    # the second pass ran for exactly the rows with a string beyond the cap (for Levenshtein: beyond 1 024 scalar values too)
    blen = lambda x: len(x.encode("utf-8"))
    if measure == "levenshtein":  # byte lengths beyond the cap, unless both sides are non-empty and fit in scalar values
        n_long = sum(1 for a, b in zip(A, B) if max(blen(a), blen(b)) > 1024 and
                     not (a and b and max(blen(a), blen(b)) <= 4096 and max(len(a), len(b)) <= 1024))
    else:
        n_long = sum(1 for a, b in zip(A, B) if max(blen(a), blen(b)) > 1024)
    assert ctx.last_long_rows == n_long, (ctx.last_long_rows, n_long)
    # and the context keeps working afterwards
    got = gpu(S, ctx, measure, A[:50], B[:50])
    assert_bit_exact(got, O.batch_strings(measure, A[:50], B[:50]), A[:50], B[:50], measure)


def test_very_long_levenshtein_stripes(S, ctx):
    """Levenshtein beyond 1024 bytes: the striped block kernel (patterns of more than 2048 scalar values take several
    stripes), ASCII / one script / mixed scripts (7, 11 and 16 planes) / astral values (anti-diagonal fallback)."""
    import random
    rng = random.Random(43)
    cyr = "".join(chr(c) for c in range(0x430, 0x450))
    cjk = "".join(chr(c) for c in range(0x4E00, 0x4E80)) + cyr + gen.ASCII_LOWER
    specs = [(1025, 1030, gen.ASCII_LOWER), (2047, 2048, gen.ASCII_LOWER), (2049, 2049, "ab"), (4100, 4500, gen.ASCII_LOWER),
             (6200, 2100, gen.ASCII_LOWER + "XYZ ,."), (700, 700, cyr), (2100, 2300, cyr), (1500, 900, cyr + gen.ASCII_LOWER),
             (2500, 2500, cjk), (33, 5000, cyr), (1200, 1300, cyr + "\U0001F600"), (4097, 1, "a"), (3000, 3000, "a")]
    A, B = [], []
    for la, lb, alpha in specs:
        a = "".join(rng.choice(alpha) for _ in range(la))
        b = gen.edit(rng, a, alpha, 7) if la == lb else "".join(rng.choice(alpha) for _ in range(lb))
        A.append(a)
        B.append(b)
    got = gpu(S, ctx, "levenshtein", A, B)
    assert_bit_exact(got, O.batch_strings("levenshtein", A, B, 8), A, B, "levenshtein")
    got = gpu(S, ctx, "levenshtein", B, A)
    assert_bit_exact(got, O.batch_strings("levenshtein", B, A, 8), B, A, "levenshtein")


@pytest.mark.parametrize("measure", ["jaccard", "sorensen_dice"])
def test_multiset_intersection_hash_many_distinct_values(S, ctx, measure):
    """Non-ASCII multiset intersection (hash table of the shorter string's scalar values): few and many distinct values,
    heavy repeats, astral values, both the LDS (<= 1024 bytes) and the workspace (longer) tables."""
    import random
    rng = random.Random(47)
    cjk = "".join(chr(c) for c in range(0x4E00, 0x4E00 + 1500))
    few = "\u0430\u0431\u0432"
    astral = "".join(chr(c) for c in range(0x1F600, 0x1F640)) + "ab\u00e9"
    A, B = [], []
    for la, lb, alpha in [(300, 320, cjk), (341, 341, cjk), (200, 30, few), (500, 510, few), (120, 100, astral), (64, 256, astral),
                          (1500, 1400, cjk), (3000, 2600, cjk), (2500, 40, few), (5000, 4000, few), (1200, 1300, astral)]:
        a = "".join(rng.choice(alpha) for _ in range(la))
        b = "".join(rng.choice(alpha) for _ in range(lb)) if rng.random() < 0.5 else gen.edit(rng, a, alpha, 9)[:max(lb, 1)]
        A.append(a)
        B.append(b)
    got = gpu(S, ctx, measure, A, B)
    assert_bit_exact(got, O.batch_strings(measure, A, B, 8), A, B, measure)


def test_shape_mismatch(S, ctx):
    with pytest.raises(S.ShapeMismatch, match="Inputs must have the same length, or one of them must be a Utf8 literal."):
        gpu(S, ctx, "jaro", ["a", "b"], ["a", "b", "c"])


def test_nulls_propagate(S):
    out = S.levenshtein(["phillips", None, "x"], ["philips", "a", None])
    assert out[0] == 0.875 and np.isnan(out[1]) and np.isnan(out[2])


def test_anchor_bits(S):
    bits = lambda x: struct.unpack("<Q", struct.pack("<d", float(x)))[0]
    a, b = ["phillips"], ["philips"]
    assert bits(S.levenshtein(a, b)[0]) == 0x3FEC000000000000
    assert bits(S.jaro(a, b)[0]) == 0x3FEEAAAAAAAAAAAB
    assert bits(S.jaro_winkler(a, b)[0]) == 0x3FEF333333333333
    assert bits(S.jaccard(a, b)[0]) == 0x3FEC000000000000
    assert bits(S.sorensen_dice(a, b)[0]) == 0x3FEDDDDDDDDDDDDE


def test_fused_all_measures_equals_single_measure_kernels(S):
    """strsim_pairs_device_all: five outputs from one pass == the five single-measure calls, bit for bit
    (mixed frame: lane path, wide path, wave path, a > 1024-byte row)."""
    import torch
    A, B = gen.pairs(71, 30000, gen.ASCII_LOWER, 0, 32)
    A2, B2 = gen.pairs(72, 3000, gen.ASCII_LOWER, 20, 128)
    A3, B3 = gen.pairs(73, 500, gen.MIXED, 0, 60)
    A, B = A + A2 + A3 + ["x" * 1500], B + B2 + B3 + ["x" * 700 + "y" * 700]
    ao, av = S.pack_strings(A)
    bo, bv = S.pack_strings(B)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    cols = (t(ao, np.int32), t(np.concatenate([av, pad]), np.uint8), t(bo, np.int32), t(np.concatenate([bv, pad]), np.uint8))
    torch.cuda.synchronize()  # the uploads are complete; the context runs on its own stream and is waited for below
    with S.Context(0) as ctx:
        outs = ctx.pairs_device_all(*cols)
        ctx.synchronize()
        torch.cuda.synchronize()
        for m, o in zip(S.MEASURES, outs):
            exp = O.batch_strings(m, A, B, 8)
            assert_bit_exact(o.cpu().numpy(), exp, A, B, "fused " + m)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_host_calls_in_place_and_copied(S, measure, monkeypatch):
    """strsim_pairs_host: small calls run in place on pinned host memory, larger ones through H2D/D2H copies; both paths on
    the same mixed frame (every row class, a > 1024-byte row), with offsets that do not start at 0, and a literal side."""
    A, B = gen.pairs(91, 700, gen.ASCII_LOWER, 0, 32)
    A2, B2 = gen.pairs(92, 100, gen.MIXED, 0, 200)
    A3, B3 = gen.pairs(93, 30, gen.ASCII_LOWER, 100, 900)
    A = ["x" * 1500] + A + A2 + A3
    B = ["x" * 700 + "y" * 700] + B + B2 + B3
    exp = O.batch_strings(measure, A, B, 8)
    ao, av = S.pack_strings(A)
    bo, bv = S.pack_strings(B)
    ao7, av7 = (ao + 7).astype(ao.dtype), np.concatenate([np.full(7, 0x7A, np.uint8), av])  # base 7: 7 bytes nobody owns
    # a base far larger than the offset arrays themselves (a slice deep inside a big Arrow column): 1 MiB of foreign bytes
    big = 1 << 20
    aoB, avB = (ao + big).astype(ao.dtype), np.concatenate([np.full(big, 0x7A, np.uint8), av])
    boB, bvB = (bo + big).astype(bo.dtype), np.concatenate([np.full(big, 0x51, np.uint8), bv])
    lo, lv = S.pack_strings(["phillips"])
    exp_lit = O.batch_strings(measure, A, ["phillips"] * len(A), 8)
    for direct in ("65536", "0"):
        monkeypatch.setenv("STRSIM_HOST_DIRECT_ROWS", direct)
        with S.Context(0) as ctx:
            assert_bit_exact(ctx.pairs_host(measure, ao, av, bo, bv), exp, A, B, "host rows<=%s" % direct)
            assert_bit_exact(ctx.pairs_host(measure, ao7, av7, bo, bv), exp, A, B, "host base 7, rows<=%s" % direct)
            assert_bit_exact(ctx.pairs_host(measure, aoB, avB, boB, bvB), exp, A, B, "host base 2^20, rows<=%s" % direct)
            assert_bit_exact(ctx.pairs_host(measure, ao, av, lo, lv), exp_lit, A, ["phillips"] * len(A), "host literal")
            for n in (1, 2, 65):
                o1, v1 = S.pack_strings(A[:n])
                o2, v2 = S.pack_strings(B[:n])
                assert_bit_exact(ctx.pairs_host(measure, o1, v1, o2, v2), exp[:n], A[:n], B[:n], "host n=%d" % n)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_transport_codec_round_trip(S, measure):
    """16-bit codec: decode(encode(x)) == x bit for bit; rows with longer strings travel as exceptions."""
    import torch
    A, B = gen.pairs(81, 50000, gen.ASCII_LOWER, 0, 32, max_bytes=32)
    A2, B2 = gen.pairs(82, 300, gen.ASCII_LOWER, 40, 120)   # outside the max_chars = 32 table (mostly)
    A, B = A + A2, B + B2
    ao, av = S.pack_strings(A)
    bo, bv = S.pack_strings(B)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    with S.Context(0) as ctx:  # its own stream: every hand-over to torch below goes through ctx.synchronize() / torch.cuda.synchronize()
        vals = ctx.pairs_device(measure, t(ao, np.int32), t(np.concatenate([av, pad]), np.uint8), t(bo, np.int32),
                                t(np.concatenate([bv, pad]), np.uint8))
        ctx.synchronize()
        codec = S.Codec(ctx, measure, 32)
        assert 300 < codec.entries < 65535
        codes = codec.encode(vals)
        out = torch.full_like(vals, -1.0)
        torch.cuda.synchronize()  # the context runs on its own stream (torch's default stream has handle 0)
        codec.decode(codes, out)
        ctx.synchronize()
        torch.cuda.synchronize()
        nexc = int(codec.exc_count.item())
        exc = codes == -1  # 0xFFFF
        assert int(exc.sum().item()) == nexc and 0 < nexc <= 300
        assert bool((out[:50000].view(torch.int64) == vals[:50000].view(torch.int64)).all())
        assert bool((out[exc] == -1.0).all())
        codec.patch(out, 0, codec.exc_rows, codec.exc_vals, nexc)
        ctx.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int64), vals.view(torch.int64))
        exp = O.batch_strings(measure, A, B, 8)
        assert_bit_exact(out.cpu().numpy(), exp, A, B, "codec " + measure)
        # the packed transport of the same codes: bits per row by the table size, same exceptions, same round trip
        bits = codec.bits
        assert (1 << bits) > codec.entries >= (1 << (bits - 1))
        for n_rows in (len(A), 50000, 7, 1):
            v = vals[:n_rows]
            words = codec.encode_packed(v)
            assert words.numel() == -(-n_rows // (64 // bits))
            out2 = torch.full_like(v, -1.0)
            torch.cuda.synchronize()  # the context runs on its own stream (torch's default stream has handle 0)
            codec.decode_packed(words, n_rows, out2)
            ctx.synchronize()
            torch.cuda.synchronize()
            nexc2 = int(codec.exc_count.item())
            assert int((out2 == -1.0).sum().item()) == nexc2
            codec.patch(out2, 0, codec.exc_rows, codec.exc_vals, nexc2)
            ctx.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(out2.view(torch.int64), v.view(torch.int64))
        codec.close()


def test_transport_codec_rejects_oversized_tables(S):
    with S.Context(0) as ctx:
        with pytest.raises(S.StrsimError, match="do not fit 16-bit"):
            S.Codec(ctx, "jaro_winkler", 128)
        c = S.Codec(ctx, "levenshtein", 128)
        assert c.entries > 1000
        c.close()


@pytest.mark.parametrize("rows", [0, 1, 15, 16, 17, 4095, 4096, 4097, 65536 + 5, 3_000_001])
def test_offsets_from_lengths(S, ctx, rows):
    """strsim_offsets_from_lengths: what the plugin ships instead of u32 offsets when every string is <= 255 bytes."""
    import torch
    rng = np.random.default_rng(rows + 1)
    lens = rng.integers(0, 256, rows, dtype=np.uint8) if rows % 2 else rng.integers(0, 33, rows, dtype=np.uint8)
    d = torch.from_numpy(lens).cuda()
    off = ctx.offsets_from_lengths(d)
    ctx.synchronize()
    torch.cuda.synchronize()
    exp = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))]).astype(np.uint32)
    assert np.array_equal(off.cpu().numpy().view(np.uint32), exp)


def _retire_frame(S, rows=30_000):
    import torch
    A, B = gen.pairs(77, rows, gen.ASCII_LOWER, 0, 40)
    exp = O.batch_strings("levenshtein", A, B, 8)
    oa, va = S.pack_strings(A)
    ob, vb = S.pack_strings(B)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    args = (t(oa, np.int32), t(np.concatenate([va, pad]), np.uint8), t(ob, np.int32), t(np.concatenate([vb, pad]), np.uint8))
    torch.cuda.synchronize()
    return args, exp


def test_calls_in_flight_are_retired_one_at_a_time(S):
    """strsim_ctx_retire_oldest: three calls enqueued back to back, each retired behind an event of the caller's own --
    recorded on the CONTEXT'S stream (a real torch stream handed to the context; everything torch does here runs on it)."""
    import torch
    args, exp = _retire_frame(S)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), S.Context(0, stream=s.cuda_stream) as ctx:
        assert ctx.stream == s.cuda_stream
        outs, evs = [], []
        for _ in range(3):
            outs.append(ctx.pairs_device("levenshtein", *args))
            ev = torch.cuda.Event()
            ev.record(s)
            evs.append(ev)
        for k in range(3):
            evs[k].synchronize()
            ctx.retire_oldest()
            got = outs[k].cpu().numpy()  # (a copy on s, the current stream)
            assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), k
        ctx.synchronize()  # nothing left to retire
    torch.cuda.synchronize()


def test_retire_oldest_refuses_a_call_that_has_not_completed(S):
    """The caller's promise is checked: with the stream held up behind the call's kernels' predecessor, retire_oldest()
    returns STRSIM_ERR_ARG and leaves the call pending; once the stream has drained the same call retires and is right."""
    import torch
    args, exp = _retire_frame(S, 5_000)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), S.Context(0, stream=s.cuda_stream) as ctx:
        torch.cuda._sleep(400_000_000)  # ~0.2 s of GPU spinning on s, in front of the call
        out = ctx.pairs_device("levenshtein", *args)
        with pytest.raises(S.StrsimError, match="has not completed"):
            ctx.retire_oldest()
        s.synchronize()
        ctx.retire_oldest()       # now it has
        ctx.retire_oldest()       # and nothing is pending: a no-op
        got = out.cpu().numpy()
        assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
        ctx.synchronize()
    torch.cuda.synchronize()


def test_default_stream_handle_is_refused(S):
    """Handle 0 would silently become an own stream inside the C ABI (NULL = create one): the Python layer says so instead."""
    with pytest.raises(ValueError, match="default stream"):
        S.Context(0, stream=0)


def _mixed_frame(S, slow):
    """30 000 short ASCII rows + `slow` rows for the kernels behind the lane kernel (33..128 bytes, non-ASCII, long)."""
    import torch
    A, B = gen.pairs(901, 30_000, gen.ASCII_LOWER, 0, 32, max_bytes=32)
    if slow:
        A2, B2 = gen.pairs(902, slow, gen.ASCII_LOWER, 40, 120)
        A3, B3 = gen.pairs(903, max(slow // 4, 1), gen.MIXED, 0, 60)
        pos = len(A) // 3
        A, B = A[:pos] + A2 + A3 + A[pos:], B[:pos] + B2 + B3 + B[pos:]
    oa, va = S.pack_strings(A)
    ob, vb = S.pack_strings(B)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    cols = (t(oa, np.int32), t(np.concatenate([va, pad]), np.uint8), t(ob, np.int32), t(np.concatenate([vb, pad]), np.uint8))
    torch.cuda.synchronize()
    return A, B, cols


@pytest.mark.parametrize("measure", O.MEASURES)
def test_one_launch_calls_and_deferred_slow_rows(S, measure):
    """A call on a context whose last call left no slow rows is ONE launch; if it does hold slow rows they are finished when
    the call is retired (last_late_rows says so) and the following calls enqueue all five operations up front again; a clean
    call brings the context back to one launch.  Frames with 0, 1 and many slow rows, every result against the oracle."""
    frames = {k: _mixed_frame(S, k) for k in (0, 1, 700)}
    exp = {k: O.batch_strings(measure, f[0], f[1], 8) for k, f in frames.items()}
    with S.Context(0, one_launch=True) as ctx:
        def call(k):
            before = ctx.enqueued_ops
            out = ctx.pairs_device(measure, *frames[k][2])
            ops = ctx.enqueued_ops - before
            ctx.synchronize()
            assert_bit_exact(out.cpu().numpy(), exp[k], frames[k][0], frames[k][1], "%s, %d slow rows" % (measure, k))
            return ops, ctx.last_late_rows
        assert call(0) == (1, 0)                    # a fresh context expects no slow rows: one launch, nothing late
        ops, late = call(1)                         # ... wrongly, this time: the slow row is finished at synchronize()
        assert ops == 1 and late >= 1
        assert call(700) == (5, 0)                  # now the whole chain goes up front
        assert call(0) == (5, 0)                    # (the prediction follows the call retired LAST)
        assert call(0) == (1, 0)
        ops, late = call(700)
        assert ops == 1 and late >= 700
        # several mispredicted calls in flight at once (more than the context has mask buffers), retired by one synchronize()
        ctx.pairs_device(measure, *frames[0][2])
        ctx.synchronize()
        outs = [ctx.pairs_device(measure, *frames[k][2]) for k in (700, 1, 700, 0, 700, 1, 700)]
        ctx.synchronize()
        for o, k in zip(outs, (700, 1, 700, 0, 700, 1, 700)):
            assert_bit_exact(o.cpu().numpy(), exp[k], frames[k][0], frames[k][1], "%s, in flight, %d slow rows" % (measure, k))


def test_ring_wrap_with_a_slow_row_call_and_an_eager_small_call(S):
    """40 one-launch calls enqueued back to back without a single retirement -- more than the context's ring of 32 status
    slots -- one of them with slow rows.  At the wrap the library retires what is pending inside strsim_pairs_device (documented);
    what that finished late is carried to the caller's next retirement, and every result is the oracle's."""
    order = [0] * 5 + [700] + [0] * 34
    frames = {k: _mixed_frame(S, k) for k in (0, 700)}
    exp = {k: O.batch_strings("jaro_winkler", f[0], f[1], 8) for k, f in frames.items()}
    with S.Context(0, one_launch=True) as ctx:
        outs = [ctx.pairs_device("jaro_winkler", *frames[k][2]) for k in order]
        ctx.synchronize()
        assert ctx.last_late_rows >= 700  # the sixth call's slow rows were finished at the wrap: reported with THIS retirement
        for o, k in zip(outs, order):
            assert_bit_exact(o.cpu().numpy(), exp[k], frames[k][0], frames[k][1], "ring wrap, %d slow rows" % k)
        # an eager small call that leaves nothing pending does not wipe what pending calls still have to report
        ctx.pairs_device("jaro", *frames[700][2])  # (pending, one launch, slow rows inside)
        A, B = frames[0][0][:100], frames[0][1][:100]
        oa, va = S.pack_strings(A)
        ob, vb = S.pack_strings(B)
        got = ctx.pairs_host("jaro", oa, va, ob, vb)  # the in-place host path: strsim_pairs_device_small + synchronize
        assert_bit_exact(got, O.batch_strings("jaro", A, B, 2), A, B, "small call beside a pending one")
        # (pairs_host synchronised the context: the pending call was retired there and its slow rows were reported, not wiped)
        assert ctx.last_late_rows >= 700


def test_failed_retirement_at_the_ring_wrap_is_reported_as_an_earlier_call(S, monkeypatch):
    """STRSIM_ERR_EARLIER_CALL (include/strsim_amd.h, ABI 1.5): 32 one-launch calls in flight, the 33rd has to retire them inside
    strsim_pairs_device and one of those retirements fails (fault injection: STRSIM_FAULT_RETIRE_AT, read at context creation).
    The 33rd call gets status 8 with the earlier call's message inside its own, was NOT enqueued (its output buffer is
    untouched, no operation was counted), the calls retired before the fault are right, and the context goes on working."""
    import torch
    frames = {k: _mixed_frame(S, k) for k in (0, 40)}
    exp = {k: O.batch_strings("levenshtein", f[0], f[1], 8) for k, f in frames.items()}
    monkeypatch.setenv("STRSIM_FAULT_RETIRE_AT", "5")
    with S.Context(0, one_launch=True) as ctx:
        monkeypatch.delenv("STRSIM_FAULT_RETIRE_AT")  # (read once, at creation)
        outs = [ctx.pairs_device("levenshtein", *frames[0][2]) for _ in range(32)]
        sentinel = torch.full((len(frames[0][0]),), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ops = ctx.enqueued_ops
        with pytest.raises(S.StrsimError) as ei:
            ctx.pairs_device("levenshtein", *frames[0][2], out=sentinel)
        assert ei.value.code == 8 and S.STATUS[8] == "ERR_EARLIER_CALL"
        assert "retiring earlier pending calls" in ei.value.message and "not enqueued" in ei.value.message
        assert "fault injected at retirement 5" in ei.value.message  # the earlier call's own message travels inside
        assert ctx.enqueued_ops == ops
        ctx.synchronize()  # nothing is pending any more: the failed drain retired (dropped) every slot
        torch.cuda.synchronize()
        assert bool((sentinel == -7.0).all())
        for k in range(4):  # retired before the fault; (clean frames: the lane kernel finished every row of the others too)
            assert_bit_exact(outs[k].cpu().numpy(), exp[0], frames[0][0], frames[0][1], "before the fault, call %d" % k)
        # the context is usable: a frame with slow rows and a clean one, both modes of enqueueing
        for k in (40, 0, 0, 40):
            out = ctx.pairs_device("levenshtein", *frames[k][2])
            ctx.synchronize()
            assert_bit_exact(out.cpu().numpy(), exp[k], frames[k][0], frames[k][1], "after the fault, %d slow rows" % k)


def test_failed_retirement_is_reported_by_synchronize_itself(S, monkeypatch):
    """The same fault outside a wrap is the retiring call's own error (STRSIM_ERR_INTERNAL from strsim_ctx_synchronize /
    strsim_ctx_retire_oldest), every slot is released, and the context goes on working."""
    frames = {k: _mixed_frame(S, k) for k in (0, 40)}
    exp = {k: O.batch_strings("jaccard", f[0], f[1], 8) for k, f in frames.items()}
    monkeypatch.setenv("STRSIM_FAULT_RETIRE_AT", "2")
    with S.Context(0, one_launch=True) as ctx:
        for _ in range(3):
            ctx.pairs_device("jaccard", *frames[0][2])
        with pytest.raises(S.StrsimError) as ei:
            ctx.synchronize()
        assert ei.value.code == 7 and "fault injected at retirement 2" in ei.value.message
        ctx.synchronize()  # nothing left
        for k in (40, 0):
            out = ctx.pairs_device("jaccard", *frames[k][2])
            ctx.synchronize()
            assert_bit_exact(out.cpu().numpy(), exp[k], frames[k][0], frames[k][1], "after the fault, %d slow rows" % k)


def test_pipelined_caller_with_early_copies_and_retire_oldest(S):
    """The documented pattern of a pipelined caller in one-launch mode -- own event behind each call, an EARLY device-to-host
    copy behind it, retire_oldest() in order, copy again when last_late_rows says a pass ran late -- with seven calls in flight:
    the library never retires a call the caller has not asked it
    to, so the k-th retire_oldest() is the k-th call, and every call whose slow rows were finished late says so."""
    import torch
    order = (0, 700, 1, 0, 700, 1, 700)
    frames = {k: _mixed_frame(S, k) for k in set(order)}
    exp = {k: O.batch_strings("jaro", f[0], f[1], 8) for k, f in frames.items()}
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), S.Context(0, stream=s.cuda_stream, one_launch=True) as ctx:
        outs, early, evs, ops = [], [], [], []
        for k in order:
            before = ctx.enqueued_ops
            outs.append(ctx.pairs_device("jaro", *frames[k][2]))
            ops.append(ctx.enqueued_ops - before)
            host = torch.empty(outs[-1].shape, dtype=torch.float64).pin_memory()
            host.copy_(outs[-1], non_blocking=True)  # the early copy, on s behind the call
            early.append(host)
            ev = torch.cuda.Event()
            ev.record(s)
            evs.append(ev)
        assert ops == [1] * len(order)  # every call is one launch: each owns its mask until the caller retires it
        late_calls = 0
        for i, k in enumerate(order):
            evs[i].synchronize()
            ctx.retire_oldest()
            late = ctx.last_late_rows
            if ops[i] == 1 and k:
                assert late >= k, (i, k, late)   # this call's slow rows ran at ITS retirement ...
            else:
                assert late == 0, (i, k, late)   # ... and nobody else's are reported here
            if late:
                late_calls += 1
                early[i].copy_(outs[i], non_blocking=True)
                s.synchronize()
            A, B = frames[k][0], frames[k][1]
            assert_bit_exact(early[i].numpy(), exp[k], A, B, "pipelined caller, call %d (%d slow rows)" % (i, k))
        assert late_calls >= 2
        ctx.retire_oldest()  # nothing pending: a no-op
        ctx.synchronize()
    torch.cuda.synchronize()


def test_fused_call_is_one_launch_without_slow_rows(S):
    A, B, cols = _mixed_frame(S, 0)
    A2, B2, cols2 = _mixed_frame(S, 50)
    with S.Context(0, one_launch=True) as ctx:
        before = ctx.enqueued_ops
        outs = ctx.pairs_device_all(*cols)
        assert ctx.enqueued_ops - before == 1
        ctx.synchronize()
        for m, o in zip(S.MEASURES, outs):
            assert_bit_exact(o.cpu().numpy(), O.batch_strings(m, A, B, 8), A, B, "fused, one launch, " + m)
        outs = ctx.pairs_device_all(*cols2)  # mispredicted: the slow rows of all five measures at synchronize()
        assert ctx.enqueued_ops - before == 2
        ctx.synchronize()
        assert ctx.last_late_rows >= 50
        for m, o in zip(S.MEASURES, outs):
            assert_bit_exact(o.cpu().numpy(), O.batch_strings(m, A2, B2, 8), A2, B2, "fused, deferred, " + m)


def test_stream_ordered_context_completes_in_stream_order(S):
    """The default mode of a context (set_stream_ordered(True), ABI 1.4): the results of a frame WITH slow rows are complete for work enqueued behind the call on the
    context's stream -- no retire in between (what bench.py's gather and any torch consumer on that stream rely on)."""
    import torch
    A, B, cols = _mixed_frame(S, 300)
    exp = O.batch_strings("jaro_winkler", A, B, 8)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), S.Context(0, stream=s.cuda_stream) as ctx:
        before = ctx.enqueued_ops
        out = ctx.pairs_device("jaro_winkler", *cols)
        assert ctx.enqueued_ops - before == 5
        got = out.cpu().numpy()  # a copy on s, behind the kernels
        assert_bit_exact(got, exp, A, B, "stream-ordered")
        ctx.synchronize()
        assert ctx.last_late_rows == 0
    torch.cuda.synchronize()


def test_compact_segments(S, ctx):
    """strsim_compact_segments: the device side of the plugin's one-pass packer -- segments at arbitrary byte offsets and of
    arbitrary byte lengths (unaligned heads and tails) moved to their final places in one launch."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, 6_000_000, dtype=np.uint8)
    sizes = [0, 1, 15, 16, 17, 4095, 1_000_003, 999_999, 123_457, 64, 2_000_001]
    src_off, dst_off, pos, dpos = [], [], 5, 0
    for sz in sizes:
        src_off.append(pos)
        dst_off.append(dpos)
        pos += sz + int(rng.integers(0, 5000))
        dpos += sz
    assert pos <= src.size
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dpos + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    arr = lambda v: np.asarray(v, dtype=np.uint64)
    so, do, by = arr(src_off), arr(dst_off), arr(sizes)
    rc = S.lib().strsim_compact_segments(ctx._h, d_src.data_ptr(), d_dst.data_ptr(), so.ctypes.data, do.ctypes.data, by.ctypes.data, len(sizes))
    assert rc == 0
    ctx.synchronize()
    torch.cuda.synchronize()
    exp = np.concatenate([src[o:o + sz] for o, sz in zip(src_off, sizes)])
    got = d_dst.cpu().numpy()
    assert np.array_equal(got[:dpos], exp) and not got[dpos:].any()
    assert S.lib().strsim_compact_segments(ctx._h, d_src.data_ptr(), d_dst.data_ptr(), so.ctypes.data, do.ctypes.data, by.ctypes.data, 33) != 0


def test_offsets_from_lengths_refuses_rows_that_could_wrap(S, ctx):
    import torch
    rows = 16_843_010  # one more than floor((2^32 - 1) / 255)
    d = torch.zeros(rows, dtype=torch.uint8, device="cuda")
    with pytest.raises(S.StrsimError, match="at most"):
        ctx.offsets_from_lengths(d)


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("side", ["a", "b"])
def test_literal_calls_are_one_launch_too(S, measure, side):
    """A column of short ASCII strings against a literal is k_lane_lit + a one-thread kernel that publishes what it left (two
    launches instead of six) on a context that expects no slow rows -- on either side, for every measure ([r5] Jaro / Jaro-Winkler
    with a literal b too: they are symmetric); a literal the lane kernels cannot take (longer than 32 bytes) is finished at
    retirement."""
    import torch
    A, B, cols = _mixed_frame(S, 0)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    for lit in ("phillips", "q" * 40):
        lo, lv = S.pack_strings([lit])
        lcols = (t(lo, np.int32), t(np.concatenate([lv, pad]), np.uint8))
        torch.cuda.synchronize()
        with S.Context(0, one_launch=True) as ctx:
            before = ctx.enqueued_ops
            out = ctx.pairs_device(measure, *(lcols + cols[:2] if side == "a" else cols[:2] + lcols))
            assert ctx.enqueued_ops - before == 2
            ctx.synchronize()
            assert (ctx.last_late_rows == 0) == (len(lit) <= 32)
            exp = O.batch_strings(measure, [lit] * len(A) if side == "a" else A, A if side == "a" else [lit] * len(A), 8)
            assert_bit_exact(out.cpu().numpy(), exp, [lit] * len(A) if side == "a" else A, A if side == "a" else [lit] * len(A),
                             "literal %s, %s" % (side, measure))


@pytest.mark.parametrize("packed", [True, False])
def test_gathered_segments_decode_in_one_launch(S, packed):
    """strsim_codec_decode_gathered (the root's side of the gather): three ranks' segments -- codes + exception block each, the
    last shard longer, as split_offsets cuts them -- decoded by ONE launch == the per-segment decode + patch."""
    import torch
    measure = "levenshtein" if packed else "jaro"
    A, B = gen.pairs(81, 30_000, gen.ASCII_LOWER, 0, 32, max_bytes=32)
    A2, B2 = gen.pairs(82, 41, gen.ASCII_LOWER, 40, 120)  # outside the table: exceptions
    A, B = A[:10_000] + A2[:20] + A[10_000:] + A2[20:], B[:10_000] + B2[:20] + B[10_000:] + B2[20:]
    n = len(A)
    parts = S.split_offsets(n, 3)
    chunk, last = parts[0][1], parts[-1][1]
    oa, va = S.pack_strings(A)
    ob, vb = S.pack_strings(B)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    torch.cuda.synchronize()
    with S.Context(0) as ctx:
        vals = ctx.pairs_device(measure, t(oa, np.int32), t(np.concatenate([va, pad]), np.uint8), t(ob, np.int32),
                                t(np.concatenate([vb, pad]), np.uint8))
        ctx.synchronize()
        codec = S.Codec(ctx, measure, 32)
        assert (64 // codec.bits > 4) == packed
        cap = 256
        code_bytes = ((8 * codec.packed_words(last) if packed else 2 * last) + 15) & ~15
        stride = code_bytes + 16 + 12 * cap
        buf = torch.zeros(3 * stride, dtype=torch.uint8, device=dev)
        nexc = []
        for r, (off, ln) in enumerate(parts):
            seg = buf[r * stride:(r + 1) * stride]
            exc = (seg[code_bytes:code_bytes + 4].view(torch.int32), seg[code_bytes + 16:code_bytes + 16 + 4 * cap].view(torch.int32),
                   seg[code_bytes + 16 + 4 * cap:code_bytes + 16 + 12 * cap].view(torch.float64))
            if packed:
                codec.encode_packed(vals[off:off + ln], seg[:code_bytes].view(torch.int64), exc=exc)
            else:
                codec.encode(vals[off:off + ln], seg[:2 * ln].view(torch.int16), exc=exc)
            ctx.synchronize()
            nexc.append(int(exc[0].item()))
        assert sum(nexc) >= 30 and max(nexc) <= cap
        out = torch.full((n,), -1.0, dtype=torch.float64, device=dev)
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        codec.decode_gathered(buf, stride, 3, chunk, last, packed, code_bytes, cap, out, overflow)
        ctx.synchronize()
        torch.cuda.synchronize()
        assert int(overflow.item()) == 0
        assert torch.equal(out.view(torch.int64), vals.view(torch.int64))
        # a block that holds fewer exceptions than its rank had: the overflow flag, not silence
        codec.decode_gathered(buf, stride, 3, chunk, last, packed, code_bytes, 4, out, overflow)
        ctx.synchronize()
        torch.cuda.synchronize()
        assert int(overflow.item()) >= 1
        codec.close()


def test_more_calls_in_flight_than_the_ring_holds(S):
    """70 asynchronous calls before one synchronize(): the ring of 32 status slots wraps twice and the four mask buffers are
    reused again and again while clean frames and frames with slow rows alternate (so the context mispredicts every few
    calls and finishes slow rows of calls that were retired on the way, not at the final synchronize)."""
    frames = {k: _mixed_frame(S, k) for k in (0, 40)}
    exp = {k: O.batch_strings("jaccard", f[0], f[1], 8) for k, f in frames.items()}
    order = [0, 0, 40, 0, 40, 40, 0] * 10
    with S.Context(0, one_launch=True) as ctx:
        outs = [ctx.pairs_device("jaccard", *frames[k][2]) for k in order]
        ctx.synchronize()
        for o, k in zip(outs, order):
            assert_bit_exact(o.cpu().numpy(), exp[k], frames[k][0], frames[k][1], "ring wrap, %d slow rows" % k)
        assert ctx.enqueued_ops >= len(order)


def test_column_from_views(S, ctx):
    """strsim_column_from_views: Utf8View slots as an engine holds them -- inline strings, strings in the
    long-string area (any order, gaps between them), empty slots -- become offsets + packed values on the device."""
    import random
    import torch
    rng = random.Random(77)
    for n in (0, 1, 2047, 2048, 2049, 70_001):
        rows = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz\xc3\xa9") for _ in range(rng.choice((0, 1, 3, 4, 7, 8, 11, 12, 13, 16, 17, 40, 255, 300))))
                for _ in range(n)]
        views = np.zeros((max(n, 1), 16), dtype=np.uint8)
        longs = bytearray()
        order = list(range(n))
        rng.shuffle(order)  # the long strings lie in any order ...
        at = {}
        for i in order:
            if len(rows[i]) > 12:
                longs += b"\xff" * rng.randint(0, 5)  # ... with gaps
                at[i] = len(longs)
                longs += rows[i]
        for i, r in enumerate(rows):
            views[i, :4] = np.frombuffer(np.uint32(len(r)).tobytes(), dtype=np.uint8)
            if len(r) <= 12:
                views[i, 4:4 + len(r)] = np.frombuffer(r, dtype=np.uint8)
            else:
                views[i, 4:8] = np.frombuffer(r[:4], dtype=np.uint8)
                views[i, 8:12] = 0xEE  # (the buffer index is not looked at)
                views[i, 12:16] = np.frombuffer(np.uint32(at[i]).tobytes(), dtype=np.uint8)
        dev = torch.device("cuda", 0)
        dv = torch.from_numpy(views[:n].reshape(-1).copy() if n else np.zeros(16, dtype=np.uint8)).to(dev)
        dl = torch.from_numpy(np.frombuffer(bytes(longs) + b"\0" * 16, dtype=np.uint8).copy()).to(dev)
        total = sum(len(r) for r in rows)
        torch.cuda.synchronize()
        if n == 0:
            dv = dv[:0]
        eo = np.zeros(n + 1, dtype=np.uint32)
        eo[1:] = np.cumsum([len(r) for r in rows], dtype=np.uint64).astype(np.uint32)
        for bounded in (True, False):  # (ABI 1.5: the extents stated / the 1.4 call)
            res = ctx.column_from_views(dv, dl, total, bounded=bounded)
            ctx.synchronize()
            off, val = res[0], res[1]
            assert (off.cpu().numpy().view(np.uint32) == eo).all(), n
            assert bytes(val.cpu().numpy()[:total]) == b"".join(rows), n
            if bounded:
                assert int(res[2].item()) == 0


def test_column_from_views_malformed_slots_cannot_fault(S, ctx):
    """a slot whose string lies outside the shipped long-string bytes, a NULL long-string buffer with long slots, a slot
    with an absurd length (the running offset leaves 32 bits): nothing is read or written outside the stated extents -- the bad
    rows are skipped and counted, the good rows in front of them arrive."""
    import torch
    dev = torch.device("cuda", 0)

    def view(length, inline=b"", at=0):
        v = np.zeros(16, dtype=np.uint8)
        v[:4] = np.frombuffer(np.uint32(length).tobytes(), dtype=np.uint8)
        if length <= 12:
            v[4:4 + len(inline)] = np.frombuffer(inline, dtype=np.uint8)
        else:
            v[12:16] = np.frombuffer(np.uint32(at).tobytes(), dtype=np.uint8)
        return v

    longs = torch.from_numpy(np.frombuffer(b"0123456789abcdefghijklmnopqrstuvwxyz", dtype=np.uint8).copy()).to(dev)
    good = [view(3, b"abc"), view(20, at=4), view(0)]
    # (1) an offset far outside the 36 shipped bytes; (2) a string that starts inside and ends outside
    for bad_view in (view(20, at=1 << 30), view(30, at=20)):
        vs = np.concatenate(good + [bad_view, view(2, b"zz")])
        off, val, bad = ctx.column_from_views(torch.from_numpy(vs).to(dev), longs, 3 + 20 + 0 + 30 + 2)
        ctx.synchronize()
        assert int(bad.item()) == 1
        assert bytes(val.cpu().numpy()[:23]) == b"abc" + b"456789abcdefghijklmn"
    # (3) long slots but no long-string buffer at all (both forms of the call)
    vs = np.concatenate(good)
    off, val, bad = ctx.column_from_views(torch.from_numpy(vs).to(dev), None, 23)
    ctx.synchronize()
    assert int(bad.item()) == 1 and bytes(val.cpu().numpy()[:3]) == b"abc"
    off, val = ctx.column_from_views(torch.from_numpy(vs).to(dev), None, 23, bounded=False)
    ctx.synchronize()
    assert bytes(val.cpu().numpy()[:3]) == b"abc"
    # (4) absurd lengths: three slots of 2^31 bytes push the running offset past 2^32 -- every row behind them is out of range
    vs = np.concatenate([view(3, b"abc")] + [view(1 << 31, at=0)] * 3 + [view(2, b"zz")] * 3000)
    off, val, bad = ctx.column_from_views(torch.from_numpy(vs).to(dev), longs, 4096)
    ctx.synchronize()
    assert int(bad.item()) == 3 + 3000 and bytes(val.cpu().numpy()[:3]) == b"abc"
    assert not val.cpu().numpy()[3:].any()  # nothing else was written


def test_c_abi_gather_single_rank(S, ctx):
    """strsim_gather_* (ABI 1.5): RCCL resolved at first use (no link-time dependency), a communicator of ONE rank on this GPU, the
    gather of its shard into the column on the context's stream -- the root's own shard is a device copy; more ranks need more GPUs
    (RCCL refuses two ranks on one device) and run only on a node.  Argument errors are reported, not crashed on."""
    import torch
    from strsim_amd.distributed import AbiGather
    dev = torch.device("cuda", 0)
    uid = AbiGather.unique_id()
    assert len(uid) == 128 and any(uid)
    g = AbiGather(ctx, uid, 1, 0)
    n = 100_003
    oa, va = S.pack_strings(["phillips"] * n)
    ob, vb = S.pack_strings(["philips"] * n)
    t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
    pad = np.zeros(64, dtype=np.uint8)
    shard = ctx.pairs_device("jaro_winkler", t(oa, np.int32), t(np.concatenate([va, pad]), np.uint8), t(ob, np.int32),
                             t(np.concatenate([vb, pad]), np.uint8))
    column = torch.zeros(n, dtype=torch.float64, device=dev)
    g.gather(shard, column, n, root=0)   # behind the kernels, in stream order
    ctx.synchronize()
    assert bool((column == 0.975).all()) and torch.equal(column.view(torch.int64), shard.view(torch.int64))
    with pytest.raises(S.StrsimError):
        g.gather(shard, column, n, root=3)
    with pytest.raises(S.StrsimError):
        AbiGather(ctx, uid, 2, 5)
    g.close()
