"""oracle (C) == model (Python) at 0 ulp -- two independent readings of strsim.rs:125-345 held against each other where the
reference's own vectors do not reach: arbitrary Unicode (astral planes included), strings of up to 1 500 characters, a literal on
either side.  CPU only; what remains unpinned after this is bit-level agreement with the Rust BINARY (no toolchain here)."""
import random
import struct

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import model_py as M
import oracle_lib as O


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def check(a, b):
    for m in O.MEASURES:
        got, exp = O.pair(m, a, b), M.MODEL[m](a, b)
        assert bits(got) == bits(exp), (m, a, b, got, exp)


# every plane: ASCII, Latin-1, BMP (CJK, Cyrillic), the last BMP scalar values around the surrogate gap, astral (emoji, U+10FFFF)
ALPHABETS = [
    st.characters(min_codepoint=0x61, max_codepoint=0x64),                       # tiny alphabet: many crossing matches
    st.characters(min_codepoint=0x20, max_codepoint=0x7E),
    st.characters(min_codepoint=0x80, max_codepoint=0x7FF),                      # two-byte sequences
    st.characters(min_codepoint=0x800, max_codepoint=0xFFFF, blacklist_categories=("Cs",)),
    st.characters(min_codepoint=0x10000, max_codepoint=0x10FFFF),                # four-byte sequences
    st.sampled_from("abéЖ日퟿￿\U00010000\U0001F600\U0010FFFF\x00\x7f\x80"),  # the boundaries of every UTF-8 length
    st.characters(blacklist_categories=("Cs",)),
]


@settings(max_examples=400, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.data())
def test_oracle_equals_model_on_arbitrary_unicode(data):
    alpha = data.draw(st.sampled_from(ALPHABETS))
    a = data.draw(st.text(alpha, max_size=48))
    # b: independent, or a with a few edits (the interesting Jaro / Levenshtein cases are near-matches)
    if data.draw(st.booleans()):
        b = data.draw(st.text(alpha, max_size=48))
    else:
        s = list(a)
        for _ in range(data.draw(st.integers(0, 4))):
            op = data.draw(st.integers(0, 2))
            if op == 0:
                s.insert(data.draw(st.integers(0, len(s))), data.draw(alpha))
            elif s and op == 1:
                del s[data.draw(st.integers(0, len(s) - 1))]
            elif s:
                s[data.draw(st.integers(0, len(s) - 1))] = data.draw(alpha)
        b = "".join(s)
    check(a, b)


@pytest.mark.parametrize("seed", range(6))
def test_oracle_equals_model_on_long_strings(seed):
    """Lengths up to 1 500 scalar values (the long-string DP, Jaro windows of hundreds of positions), four alphabets."""
    rng = random.Random(9000 + seed)
    alpha = ["ab", "abcdefghijklmnopqrstuvwxyz", "aé日😀𝄞", "".join(chr(c) for c in range(0x400, 0x460))][seed % 4]
    la = rng.choice([1, 33, 64, 65, 127, 128, 129, 700, 1024, 1025, 1500])
    a = "".join(rng.choice(alpha) for _ in range(la))
    if seed % 2:
        b = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 1500)))
    else:  # a near copy: a block moved, some characters dropped and replaced
        s = list(a)
        k = rng.randint(0, max(0, len(s) - 10))
        blk = s[k:k + 10]
        del s[k:k + 10]
        p = rng.randint(0, len(s))
        s[p:p] = blk
        for _ in range(rng.randint(0, 20)):
            if s:
                s[rng.randrange(len(s))] = rng.choice(alpha)
        b = "".join(s[rng.randint(0, 3):])
    check(a, b)
    check(b, a)


def test_oracle_equals_model_against_a_literal_on_either_side():
    """The broadcast of strsim.rs:61-66, :85-92: one string against every row, on the left and on the right."""
    rng = random.Random(77)
    rows = ["".join(rng.choice("abcdé日😀") for _ in range(rng.randint(0, 40))) for _ in range(300)] + ["", "phillips", "日本語"]
    for lit in ["", "a", "phillips", "日本語テキスト", "😀😀", "x" * 33, "ab" * 70]:
        for m in O.MEASURES:
            right = O.batch_strings(m, rows, [lit], 2)
            left = O.batch_strings(m, [lit], rows, 2)
            for i, r in enumerate(rows):
                assert bits(float(right[i])) == bits(M.MODEL[m](r, lit)), (m, r, lit)
                assert bits(float(left[i])) == bits(M.MODEL[m](lit, r)), (m, lit, r)


def test_model_reproduces_the_reference_vectors():
    """The second reading is held to the reference's own 1 115 vectors too (1e-8, strsim.rs:350)."""
    from golden_data import reference_vectors
    n = 0
    for m, _, a, b, exp in reference_vectors():
        assert abs(M.MODEL[m](a, b) - exp) <= 1e-8, (m, a, b)
        n += 1
    assert n == 1115
