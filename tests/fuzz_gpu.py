#!/usr/bin/env python3
"""Differential fuzz of the HIP path against the CPU oracle: random scripts x length classes x measures x literal sides.
Usage: python tests/fuzz_gpu.py [seconds] [seed].  Exits non-zero on the first mismatch (prints the row)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))  # the oracle is test infrastructure: this script lives in tests/
import numpy as np

import gen
import oracle_lib as O
import strsim_amd as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
ALPHABETS = {
    "a-z": gen.ASCII_LOWER, "ab": "ab", "ascii": "".join(chr(c) for c in range(1, 128)),
    "latin1": gen.ASCII_LOWER + "éèüñß ", "cyrillic": "".join(chr(c) for c in range(0x430, 0x450)) + " ",
    "mixed": gen.MIXED, "cjk": "".join(chr(c) for c in range(0x4E00, 0x4E40)) + "ab",
    "astral": "abé" + "".join(chr(c) for c in range(0x1F600, 0x1F610)),
}
CLASSES = [(0, 8, 20000), (0, 40, 20000), (20, 140, 6000), (100, 1100, 600), (900, 2600, 60), (0, 1100, 1500)]
ctx = S.Context(0)


def bounds_violations():
    """Lab build only (EXTRA="-DSTRSIM_LAB -DSTRSIM_BOUNDS", selected with STRSIM_AMD_LIB): the kernels' address-check records, read and
    cleared -> [(unit, hits, kernel, site, row, value, lo, hi)] of the units that saw a violation.  The product library has no such
    symbols: []."""
    import ctypes as C
    out = []
    for unit in ("kernels", "codec"):
        f = getattr(S.lib(), "strsim_debug_bounds_" + unit, None)
        if f is None:
            continue
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        rec = (C.c_ulonglong * 6)()
        assert f(rec) == 0
        if rec[0]:
            out.append((unit, rec[0], rec[1] >> 32, rec[1] & 0xFFFFFFFF, rec[2], hex(rec[3]), hex(rec[4]), hex(rec[5])))
    return out


CHECKED = hasattr(S.lib(), "strsim_debug_bounds_kernels")
t_end = time.time() + budget
rounds = 0
while time.time() < t_end:
    name = rng.choice(list(ALPHABETS))
    alpha = ALPHABETS[name]
    lo, hi, n = rng.choice(CLASSES)
    n = rng.randrange(1, n + 1)
    measure = rng.choice(O.MEASURES)
    A, B = gen.pairs(rng.randrange(1 << 30), n, alpha, lo, hi, p_edit=rng.choice([0.2, 0.5, 0.9]), p_same=0.05)
    side = rng.choice(["none", "none", "left", "right"])
    if side == "left":
        A = [A[rng.randrange(n)]]
    elif side == "right":
        B = [B[rng.randrange(n)]]
    ao, av = S.pack_strings(A)
    bo, bv = S.pack_strings(B)
    if os.environ.get("STRSIM_FUZZ_VERBOSE"):  # (a GPU fault kills the process: say what was running)
        print(f"round {rounds + 1}: {measure} {name} [{lo},{hi}] n={n} literal={side} bytes a={len(av)} b={len(bv)}", flush=True)
    got = ctx.pairs_host(measure, ao, av, bo, bv)
    viol = bounds_violations()
    if viol:
        print(f"ADDRESS OUT OF BOUNDS round {rounds + 1}: {measure} {name} [{lo},{hi}] n={n} literal={side} seed={seed}: "
              f"(unit, hits, kernel, site, row, value, lo, hi) = {viol}")
        sys.exit(2)
    A2 = A * n if len(A) == 1 else A
    B2 = B * n if len(B) == 1 else B
    exp = O.batch_strings(measure, A2, B2, 16)
    bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
    rounds += 1
    if bad.size:
        i = int(bad[0])
        print(f"MISMATCH round {rounds}: {measure} {name} [{lo},{hi}] n={n} literal={side}: {bad.size} rows; row {i}: "
              f"a={A2[i]!r} b={B2[i]!r} got={got[i]!r} exp={exp[i]!r}")
        sys.exit(1)
print(f"fuzz ok: {rounds} rounds in {budget:.0f} s (seed {seed}){', every address checked (lab build)' if CHECKED else ''}")
