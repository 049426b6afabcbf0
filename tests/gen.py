"""Seeded random string-pair generators for parity tests (pure Python, small sizes)."""
import random

ASCII_LOWER = "abcdefghijklmnopqrstuvwxyz"
MIXED = "abcdefghij" + "ABC" + " -'." + "éèüñ" + "日本語" + "😀𝄞"


def rand_string(rng, alphabet, lo, hi):
    return "".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi)))


def edit(rng, s, alphabet, k):
    s = list(s)
    for _ in range(k):
        op = rng.randint(0, 2)
        if op == 0:
            s.insert(rng.randint(0, len(s)), rng.choice(alphabet))
        elif op == 1 and s:
            del s[rng.randrange(len(s))]
        elif s:
            s[rng.randrange(len(s))] = rng.choice(alphabet)
    return "".join(s)


def pairs(seed, n, alphabet=ASCII_LOWER, lo=0, hi=32, p_edit=0.5, p_same=0.05, max_bytes=None):
    """b is an edited copy of a (p_edit), identical (p_same) or independent -- the BASELINE.md input law."""
    rng = random.Random(seed)
    A, B = [], []
    while len(A) < n:
        a = rand_string(rng, alphabet, lo, hi)
        r = rng.random()
        if r < p_edit:
            b = edit(rng, a, alphabet, rng.randint(1, 3))
        elif r < p_edit + p_same:
            b = a
        else:
            b = rand_string(rng, alphabet, lo, hi)
        if max_bytes is not None and (len(a.encode()) > max_bytes or len(b.encode()) > max_bytes):
            continue
        A.append(a)
        B.append(b)
    return A, B
