"""Host-side check of the per-lane bit-parallel cores (strsim_lane_core.h) against the oracle.

The same header is compiled into the gfx950 lane-per-pair kernels; here it is built with g++ and
driven pair by pair, so algorithmic errors surface without a GPU.  Bit-exact comparison.
"""
import ctypes as C
import os
import random
import struct
import subprocess

import pytest

import oracle_lib as O
from golden_data import reference_vectors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDIR = os.path.join(ROOT, "tests", "cpu_harness")
CSRC = os.path.join(ROOT, "polars-strsim_amd", "csrc")


@pytest.fixture(scope="module")
def harness():
    so = os.path.join(HDIR, "liblane_core_harness.so")
    srcs = [os.path.join(HDIR, "lane_core_harness.cpp"), os.path.join(CSRC, "strsim_lane_core.h"), os.path.join(HDIR, "lane_core_textbook.h"),
            os.path.join(CSRC, "strsim_lane_wide.h"), os.path.join(CSRC, "strsim_lane_sym.h"),
            os.path.join(CSRC, "strsim_lane_lut.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
                               "-I", CSRC, "-o", so, srcs[0]])
    L = C.CDLL(so)
    L.harness_lane_pair.restype = C.c_double
    L.harness_lane_pair.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.c_int]
    L.harness_lane_pair_wide.restype = C.c_double
    L.harness_lane_pair_wide.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.c_int]
    L.harness_lane_pair_wide_tp.restype = C.c_double
    L.harness_lane_pair_wide_tp.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.c_int]
    L.harness_lane_pair_sym.restype = C.c_double
    L.harness_div3_mismatches.restype = C.c_long
    L.harness_div3_mismatches.argtypes = [C.POINTER(C.c_long)]
    L.harness_set_zip_mode.restype = None
    L.harness_set_zip_mode.argtypes = [C.c_int, C.c_uint32]
    L.harness_lane_pair_sym.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int]
    L.harness_lev_snap.restype = C.c_uint32
    L.harness_lev_snap.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.harness_cores32.restype = C.c_int
    L.harness_cores32.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                  C.POINTER(C.c_uint32)]
    L.harness_utf8_decode.restype = C.c_uint32
    L.harness_utf8_decode.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]
    L.harness_check_planes.restype = C.c_int
    L.harness_check_planes.argtypes = [C.c_char_p]
    return L


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def lane(L, m, a, b, force_np=0, fill=ord("q")):
    a = a.encode() if isinstance(a, str) else a
    b = b.encode() if isinstance(b, str) else b
    return L.harness_lane_pair(O.MEASURE_ID[m], a, len(a), b, len(b), force_np, fill)


def test_bit_plane_transpose(harness):
    rng = random.Random(1)
    for _ in range(2000):
        bs = bytes(rng.randrange(256) for _ in range(32))
        assert harness.harness_check_planes(bs) == 0
    for i in range(32):
        for k in range(8):
            bs = bytearray(32)
            bs[i] = 1 << k
            assert harness.harness_check_planes(bytes(bs)) == 0


def test_lane_core_on_reference_vectors(harness):
    for m, fn, a, b, exp in reference_vectors():
        got = lane(harness, m, a, b)
        assert abs(got - exp) < 1e-8, (fn, a, b, got, exp)
        assert bits(got) == bits(O.pair(m, a, b)), (fn, a, b)


def _rand_pairs(rng, n, alphabet, maxlen=32):
    out = []
    for _ in range(n):
        la = rng.randint(0, maxlen)
        a = bytes(rng.choice(alphabet) for _ in range(la))
        r = rng.random()
        if r < 0.1:
            b = a
        elif r < 0.6:
            b = bytearray(a)
            for _ in range(rng.randint(1, 3)):
                op = rng.randint(0, 2)
                if op == 0 and len(b) < maxlen:
                    b.insert(rng.randint(0, len(b)), rng.choice(alphabet))
                elif op == 1 and len(b) > 0:
                    del b[rng.randrange(len(b))]
                elif len(b) > 0:
                    b[rng.randrange(len(b))] = rng.choice(alphabet)
            b = bytes(b)
        else:
            b = bytes(rng.choice(alphabet) for _ in range(rng.randint(0, maxlen)))
        out.append((a, b))
    return out


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("alphabet", [b"ab", b"abcdefghijklmnopqrstuvwxyz", bytes(range(1, 128))])
def test_lane_core_random_bit_exact(harness, measure, alphabet):
    rng = random.Random(hash((measure, len(alphabet))) & 0xFFFF)
    for n, (a, b) in enumerate(_rand_pairs(rng, 3000, alphabet)):
        exp = O.pair(measure, a, b)
        # planes chosen like the kernel does, with window filler from the same alphabet ...
        got = lane(harness, measure, a, b, 0, alphabet[n % len(alphabet)])
        assert bits(got) == bits(exp), (measure, a, b, got, exp)
        # ... and forced to the widest settings (zero filler = end of buffer)
        got = lane(harness, measure, a, b, 7 + (n & 1), 0)
        assert bits(got) == bits(exp), (measure, a, b, got, exp)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_lane_core_length_boundaries(harness, measure):
    rng = random.Random(7)
    for la in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32):
        for lb in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32):
            for _ in range(20):
                a = bytes(rng.choice(b"abc") for _ in range(la))
                b = bytes(rng.choice(b"abc") for _ in range(lb))
                assert bits(lane(harness, measure, a, b)) == bits(O.pair(measure, a, b)), (a, b)


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("W", [1, 2, 3, 4])
def test_wide_cores_random_bit_exact(harness, measure, W):
    rng = random.Random(1000 + W)
    for alphabet in (b"ab", b"abcdefghijklmnopqrstuvwxyz", bytes(range(1, 128))):
        for n, (a, b) in enumerate(_rand_pairs(rng, 700, alphabet, maxlen=32 * W)):
            if not a or not b:
                continue
            exp = O.pair(measure, a, b)
            for force, fill in ((0, alphabet[n % len(alphabet)]), (7, 0)):
                got = harness.harness_lane_pair_wide(O.MEASURE_ID[measure], W, a, len(a), b, len(b), force, fill)
                assert bits(got) == bits(exp), (measure, W, a, b, got, exp)


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("W", [1, 2, 3, 4])
def test_wide_cores_masks_as_wide_as_the_pattern(harness, measure, W):
    """k_lane_wide sizes its masks by the pattern alone: a Jaro row with a 100-byte a and a 10-byte b runs one-word masks
    over 100 columns (strsim.rs:200-237 walks a against b whatever their lengths)."""
    rng = random.Random(2000 + W)
    lens_a = (1, 5, 31, 32, 33, 40, 63, 64, 65, 96, 97, 127, 128)
    lens_b = sorted({1, 2, 7, 32 * W - 1, 32 * W, max(1, 32 * W - 16), max(1, 32 * (W - 1) + 1)})
    for alphabet in (b"ab", b"abcdefghijklmnopqrstuvwxyz", bytes(range(1, 128))):
        for la in lens_a:
            for lb in lens_b:
                for _ in range(4):
                    a = bytes(rng.choice(alphabet) for _ in range(la))
                    b = bytes(rng.choice(alphabet) for _ in range(lb))
                    if rng.random() < 0.5:  # related strings: matches, transpositions, shared characters
                        b = bytes(rng.choice(a) for _ in range(lb))
                    exp = O.pair(measure, a, b)
                    for force, fill in ((0, alphabet[(la + lb) % len(alphabet)]), (7, 0)):
                        got = harness.harness_lane_pair_wide_tp(O.MEASURE_ID[measure], W, a, la, b, lb, force, fill)
                        assert bits(got) == bits(exp), (measure, W, a, b, got, exp)


@pytest.mark.parametrize("measure", O.MEASURES)
def test_wide_cores_length_boundaries(harness, measure):
    rng = random.Random(3)
    lens = (1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128)
    for la in lens:
        for lb in lens:
            W = 4 if max(la, lb) > 96 else (3 if max(la, lb) > 64 else 2)
            a = bytes(rng.choice(b"abc") for _ in range(la))
            b = bytes(rng.choice(b"abc") for _ in range(lb))
            got = harness.harness_lane_pair_wide(O.MEASURE_ID[measure], W, a, la, b, lb, 0, ord("c"))
            assert bits(got) == bits(O.pair(measure, a, b)), (W, a, b)


SCRIPTS = {
    "latin1": "abcdeéèüñöçß ",
    "cyrillic": "абвгдежзийклмнопрстуфхцчшщыэюя ",
    "greek+ascii": "αβγδεζηθικλμνξοπρστυφχψω abc-",
    "cjk": "日本語中文字漢한국어テキスト",
    "mixed": "aé日я-α𝄞",  # includes an astral scalar value: not eligible for the symbol path
}


@pytest.mark.parametrize("measure", O.MEASURES)
@pytest.mark.parametrize("script", sorted(SCRIPTS))
def test_symbol_cores_bit_exact(harness, measure, script):
    """Per-lane UTF-8 decode + up-to-16-plane cores (strsim_lane_sym.h) vs the oracle on scalar values."""
    rng = random.Random(len(script) * 7 + O.MEASURE_ID[measure])
    alpha = SCRIPTS[script]
    eligible = 0
    for n in range(1500):
        la = rng.randint(1, 34)
        a = "".join(rng.choice(alpha) for _ in range(la))
        r = rng.random()
        if r < 0.1:
            b = a
        elif r < 0.6:
            bl = list(a)
            for _ in range(rng.randint(1, 3)):
                op = rng.randint(0, 2)
                if op == 0:
                    bl.insert(rng.randint(0, len(bl)), rng.choice(alpha))
                elif op == 1 and bl:
                    del bl[rng.randrange(len(bl))]
                elif bl:
                    bl[rng.randrange(len(bl))] = rng.choice(alpha)
            b = "".join(bl)
        else:
            b = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 34)))
        ab, bb = a.encode(), b.encode()
        ok = 0 < len(a) <= 32 and 0 < len(b) <= 32 and len(ab) <= 128 and len(bb) <= 128 and all(ord(c) <= 0xFFFF for c in a + b)
        for force in (0, 16):
            got = harness.harness_lane_pair_sym(O.MEASURE_ID[measure], ab, len(ab), bb, len(bb), force)
            if not ok:
                assert got == -1.0, (a, b)
            else:
                assert bits(got) == bits(O.pair(measure, a, b)), (measure, a, b, got)
        eligible += ok
    assert eligible > 300 or script == "mixed"


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]


@pytest.mark.parametrize("np_,alphabet", [(5, "abcdefghijklmnopqrstuvwxyz"), (5, "ab"), (7, "aZ09 ~!bcXY")])
def test_levenshtein_snapshot_core(harness, np_, alphabet):
    """lev_myers32_snap (the full-rate arrangement the staged kernel runs) = the textbook distance (strsim.rs:141-160), for
    every way a wave can run it: the lane's text may end anywhere in [tmin, tmax], bytes behind the strings are garbage."""
    rng = random.Random(20 + np_)
    fill = ord("q") if np_ == 5 else ord("#")
    for _ in range(3000):
        la, lb = rng.randint(1, 32), rng.randint(1, 32)
        a = "".join(rng.choice(alphabet) for _ in range(la))
        if rng.random() < 0.5:
            b = list(a)
            for _e in range(rng.randint(0, 3)):
                k = rng.randrange(len(b) + 1)
                r = rng.random()
                if r < 0.34 and b:
                    del b[min(k, len(b) - 1)]
                elif r < 0.67:
                    b.insert(k, rng.choice(alphabet))
                elif b:
                    b[min(k, len(b) - 1)] = rng.choice(alphabet)
            b = "".join(b)[:32] or rng.choice(alphabet)
        else:
            b = "".join(rng.choice(alphabet) for _ in range(lb))
        tmin = rng.randint(1, len(a))
        tmax = min(32, (rng.randint(len(a), 32) + 1) & ~1)
        got = harness.harness_lev_snap(a.encode(), len(a), b.encode(), len(b), tmin, max(tmax, len(a) + (len(a) & 1)), np_, fill)
        assert got == _edit_distance(a, b), (a, b, tmin, tmax)


def _jaro_ints(a, b):
    """(m, t) of Jaro::compute, strsim.rs:200-237: t = unequal pairs in the zip of the flagged characters, NOT halved."""
    la, lb = len(a), len(b)
    bound = max(la, lb) // 2
    bound = bound - 1 if bound > 0 else 0
    fa, fb = [False] * la, [False] * lb
    m = 0
    for i in range(la):
        lo = max(0, i - bound)
        hi = min(lb - 1, i + bound)
        for j in range(lo, hi + 1):
            if a[i] == b[j] and not fb[j]:
                fa[i] = fb[j] = True
                m += 1
                break
    xa = [a[i] for i in range(la) if fa[i]]
    xb = [b[j] for j in range(lb) if fb[j]]
    return m, sum(1 for x, y in zip(xa, xb) if x != y)


def _isect(a, b):
    from collections import Counter
    ca, cb = Counter(a), Counter(b)
    return sum(min(ca[c], cb[c]) for c in ca)


@pytest.mark.parametrize("np_,alphabet", [(5, "abcdefghijklmnopqrstuvwxyz"), (5, "ab"), (5, "abc"), (7, "aZ09 ~!bcXY")])
def test_one_loop_cores(harness, np_, alphabet):
    """lane_cores32 (one column loop, one match mask per column for the three cores -- what the staged kernels run for Jaro,
    Jaro-Winkler, Jaccard, Dice and the five-output pass) = edit distance, Jaro's (m, t) and the multiset intersection by
    their definitions, and the single-core instantiations agree with the fused one; every way a wave can run it."""
    rng = random.Random(70 + np_ + len(alphabet))
    fill = ord("q") if np_ == 5 else ord("#")
    out = (C.c_uint32 * 4)()
    for _ in range(3000):
        la, lb = rng.randint(1, 32), rng.randint(1, 32)
        a = "".join(rng.choice(alphabet) for _ in range(la))
        if rng.random() < 0.5:
            b = list(a)
            for _e in range(rng.randint(0, 3)):
                k = rng.randrange(len(b) + 1)
                r = rng.random()
                if r < 0.25 and b:
                    del b[min(k, len(b) - 1)]
                elif r < 0.5:
                    b.insert(k, rng.choice(alphabet))
                elif r < 0.75 and len(b) > 1:
                    k = min(k, len(b) - 2)
                    b[k], b[k + 1] = b[k + 1], b[k]
                elif b:
                    b[min(k, len(b) - 1)] = rng.choice(alphabet)
            b = "".join(b)[:32] or rng.choice(alphabet)
        else:
            b = "".join(rng.choice(alphabet) for _ in range(lb))
        tmin = rng.randint(1, len(a))
        tmax = max(min(32, (rng.randint(len(a), 32) + 1) & ~1), len(a) + (len(a) & 1))
        rc = harness.harness_cores32(a.encode(), len(a), b.encode(), len(b), tmin, tmax, np_, fill, out)
        assert rc == 0, (rc, a, b, tmin, tmax)
        assert out[0] == _edit_distance(a, b), (a, b, tmin, tmax)
        assert (out[1], out[2]) == _jaro_ints(a, b), (a, b, tmin, tmax, out[1], out[2])
        assert out[3] == _isect(a, b), (a, b, tmin, tmax)


@pytest.mark.parametrize("fill", [0x00, 0x61, 0xAB, 0xCD, 0xE4, 0xF0])
def test_utf8_decode_lane(harness, fill):
    """utf8_decode_lane (branch-free, four byte positions per trip) = str::chars() for valid UTF-8 of up to 128 bytes: the
    values in order, their count, OR / AND of the values, `big` iff one is beyond the BMP; whatever bytes follow the string in
    the window (continuation bytes, lead bytes of every class) are not seen."""
    rng = random.Random(900 + fill)
    pools = ["abcxyz 09", "éüñß¢", "абвгдя", "αβγω", "日本語한", "\u0800\uffff\u07ff\u0080\u007f", "a𝄞😀"]
    out = (C.c_uint16 * 132)()
    flags = (C.c_uint32 * 3)()
    for _ in range(4000):
        pool = "".join(rng.sample(pools, rng.randint(1, 3)))
        s = ""
        while True:
            c = rng.choice(pool)
            if len((s + c).encode()) > rng.choice((8, 40, 128)) or len(s) >= 128:
                break
            s += c
        b = s.encode()
        for i in range(132):
            out[i] = 0x5A5A
        cnt = harness.harness_utf8_decode(b, len(b), fill, out, flags)
        cps = [ord(ch) for ch in s]
        assert cnt == len(cps), (s, cnt)
        assert flags[0] == (1 if any(cp > 0xFFFF for cp in cps) else 0), s
        if not flags[0]:
            assert [out[i] for i in range(cnt)] == cps, s
            if cps:
                o = a = cps[0]
                for cp in cps:
                    o |= cp
                    a &= cp
                assert flags[1] & 0xFFFF == o and flags[2] == a, (s, hex(flags[1]), hex(flags[2]))


@pytest.mark.parametrize("measure", ["jaro", "jaro_winkler"])
@pytest.mark.parametrize("mode,slack", [(0, 0), (1, 0), (1, 8), (1, 128), (2, 0)])
def test_wide_jaro_zip_pass_over_b_or_over_the_matches(harness, measure, mode, slack):
    """jaro_wide's zip pass ([r5]): over the positions of b, or over the matched characters of a against the lowest flag of b not
    used yet -- the wave picks one (the text is the shorter string, so the second way wins when the pattern is much longer).  Both
    forms, the second one also as far as a LARGER m than the lane's own (the wave's maximum is some other lane's), and the
    kernel's own rule, over every width and text / pattern length class, crossing matches included (small alphabets)."""
    rng = random.Random(4200 + mode + slack)
    harness.harness_set_zip_mode(mode, slack)
    try:
        for W in (1, 2, 3, 4):
            for alphabet in (b"ab", b"abcd", b"abcdefghijklmnopqrstuvwxyz"):
                for _ in range(120):
                    lb = rng.randint(max(1, 32 * (W - 1) + 1), 32 * W)
                    la = rng.choice([rng.randint(1, 12), rng.randint(1, lb), rng.randint(1, 128)])
                    a = bytes(rng.choice(alphabet) for _ in range(la))
                    r = rng.random()
                    if r < 0.4:
                        b = bytes(rng.choice(alphabet) for _ in range(lb))
                    elif r < 0.7:
                        b = bytes(rng.choice(a) for _ in range(lb))
                    else:  # a inside b, some of it in another order
                        s_ = list(a[:lb]) + [rng.choice(alphabet) for _ in range(max(0, lb - la))]
                        for _k in range(4):
                            i, j = rng.randrange(lb), rng.randrange(lb)
                            s_[i], s_[j] = s_[j], s_[i]
                        b = bytes(s_)
                    exp = O.pair(measure, a, b)
                    for force, fill in ((0, alphabet[0]), (7, 0)):
                        got = harness.harness_lane_pair_wide_tp(O.MEASURE_ID[measure], W, a, la, b, lb, force, fill)
                        assert bits(got) == bits(exp), (measure, W, mode, slack, a, b, got, exp)
    finally:
        harness.harness_set_zip_mode(2, 0)


def test_jaro_division_by_three_is_exact(harness):
    """The table epilogues of Jaro divide by 3.0 with a multiply and one exact Newton correction (div3_exact): the same double as the
    reference's `/ 3.0` (strsim.rs:241-242) for EVERY sum they can see -- all (m, la, lb, t / 2) with lengths up to 64."""
    n = C.c_long()
    assert harness.harness_div3_mismatches(C.byref(n)) == 0
    assert n.value == 810160
