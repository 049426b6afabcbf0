"""A SECOND, independent restatement of the reference's five measures, in plain Python -- test infrastructure only.

Written from /root/reference/src/expressions/strsim.rs:125-345 itself, NOT from oracle/strsim_oracle.c: the C oracle is the
checker every GPU parity test trusts, and outside the reference's own 1 115 vectors (lowercase ASCII, <= 27 characters, 1e-8) it
was pinned by one author's reading of the Rust source alone.  This file is a second reading.  A Python `str` is a sequence of
Unicode scalar values, which is what Rust's `str::chars()` yields (strsim.rs:133,138,189,194,262-263,297,299,333,335), so the UTF-8
decoder of the C oracle is checked against CPython's; Python floats are IEEE-754 binary64 and `/`, `+`, `-`, `*` on them are the
same correctly rounded operations Rust's f64 uses (CPython never fuses a multiply-add).

Never imported by the product, by bench.py or by anything that runs on the GPU box except as a test module
(tests/test_oracle_vs_model.py: oracle == model at 0 ulp).
"""


def levenshtein(a: str, b: str) -> float:
    # strsim.rs:127-130
    if (len(a) == 0 and len(b) == 0) or a == b:
        return 1.0
    # :141-145 -- one [usize; 2] per prefix of b, first column 0..=len(b)
    matrix = [[i, 0] for i in range(len(b) + 1)]
    # :146-159
    for i, a_i in enumerate(a):
        v0, v1 = i % 2, (i + 1) % 2
        matrix[0][v1] = i + 1
        for j, b_j in enumerate(b):
            diag = matrix[j][v0] if a_i == b_j else matrix[j][v0] + 1
            matrix[j + 1][v1] = min(min(diag, matrix[j + 1][v0] + 1), matrix[j][v1] + 1)
    # :160
    return 1.0 - (float(matrix[len(b)][len(a) % 2]) / float(max(len(a), len(b))))


def jaro(a: str, b: str) -> float:
    # strsim.rs:182-186
    if (len(a) == 0 and len(b) == 0) or a == b:
        return 1.0
    if len(a) == 0 or len(b) == 0:
        return 0.0
    # :197-199
    if len(a) == 1 and len(b) == 1:
        return 1.0 if a[0] == b[0] else 0.0
    # :200 (usize arithmetic: max >= 2 here, so no underflow)
    bound = max(len(a), len(b)) // 2 - 1
    m = 0
    # :202-207
    flagged = [[False, False] for _ in range(max(len(a), len(b)))]
    # :208-219 -- `.take(b.len() + bound)` then enumerate
    for i, a_i in enumerate(a[: len(b) + bound]):
        lowerbound = 0 if bound > i else i - bound
        upperbound = min(i + bound, len(b) - 1)
        for j in range(lowerbound, upperbound + 1):  # lowerbound..=upperbound (empty when lowerbound > upperbound)
            if a_i == b[j] and not flagged[j][1]:
                m += 1
                flagged[i][0] = True
                flagged[j][1] = True
                break
    # :220-237 -- zip of the flagged positions of a and of b, in order
    ia = [i for i, f in enumerate(flagged) if f[0]]
    jb = [j for j, f in enumerate(flagged) if f[1]]
    t = sum(1 for i, j in zip(ia, jb) if a[i] != b[j])
    # :238-243
    if m == 0:
        return 0.0
    return (float(m) / float(len(a)) + float(m) / float(len(b)) + float(m - t // 2) / float(m)) / 3.0


def jaro_winkler(a: str, b: str) -> float:
    # strsim.rs:258-270
    j = jaro(a, b)
    if j > 0.7:
        shared = 0
        for c, d in list(zip(a, b))[:4]:  # .zip().take(4).take_while(==)
            if c != d:
                break
            shared += 1
        return j + (float(shared) * 0.1 * (1.0 - j))
    return j


def _counts(a: str, b: str):
    buf = {}  # HashMap<char, [usize; 2]>
    for c in a:
        buf.setdefault(c, [0, 0])[0] += 1
    for c in b:
        buf.setdefault(c, [0, 0])[1] += 1
    return buf


def jaccard(a: str, b: str) -> float:
    # strsim.rs:288-292
    if (len(a) == 0 and len(b) == 0) or a == b:
        return 1.0
    if len(a) == 0 or len(b) == 0:
        return 0.0
    # :297-306
    f0 = f1 = 0
    for v in _counts(a, b).values():
        f0 += min(v[0], v[1])
        f1 += max(v[0], v[1])
    return float(f0) / float(f1)


def sorensen_dice(a: str, b: str) -> float:
    # strsim.rs:324-328
    if (len(a) == 0 and len(b) == 0) or a == b:
        return 1.0
    if len(a) == 0 or len(b) == 0:
        return 0.0
    # :333-343 -- frac = [sum of min, sum of v0 + v1, 0]
    f = [0, 0, 0]
    for v in _counts(a, b).values():
        f[0] += min(v[0], v[1])
        f[1] += v[0]
        f[1] += v[1]
    return 2.0 * float(f[0]) / float(f[1] + f[2])


MODEL = {"levenshtein": levenshtein, "jaro": jaro, "jaro_winkler": jaro_winkler, "jaccard": jaccard, "sorensen_dice": sorensen_dice}
