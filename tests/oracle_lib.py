"""ctypes loader for the CPU parity oracle (oracle/libstrsim_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MEASURES = ["levenshtein", "jaro", "jaro_winkler", "jaccard", "sorensen_dice"]
MEASURE_ID = {m: i for i, m in enumerate(MEASURES)}

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("STRSIM_ORACLE_LIB")  # (tests/run_sanitizers.sh: a sanitizer build of the oracle)
    if not so:
        so = os.path.join(ORACLE_DIR, "libstrsim_oracle.so")
        src = os.path.join(ORACLE_DIR, "strsim_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    L = C.CDLL(so)
    L.oracle_pair.restype = C.c_double
    L.oracle_pair.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.oracle_lev_rational.restype = C.c_int
    L.oracle_lev_rational.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oracle_split_offsets.restype = None
    L.oracle_split_offsets.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    for name, ot in (("oracle_batch", C.c_void_p), ("oracle_batch_u32", C.c_void_p)):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                      C.c_void_p, C.c_int]
    _lib = L
    return L


def pair(measure, a, b):
    """Similarity of one pair; a/b are str or bytes (UTF-8)."""
    if isinstance(a, str):
        a = a.encode("utf-8")
    if isinstance(b, str):
        b = b.encode("utf-8")
    return lib().oracle_pair(MEASURE_ID[measure], a, len(a), b, len(b))


def lev_rational(a, b):
    if isinstance(a, str):
        a = a.encode("utf-8")
    if isinstance(b, str):
        b = b.encode("utf-8")
    d, n = C.c_uint64(), C.c_uint64()
    lib().oracle_lev_rational(a, len(a), b, len(b), C.byref(d), C.byref(n))
    return d.value, n.value


def split_offsets(length, n):
    out = np.zeros(2 * n, dtype=np.uint64)
    lib().oracle_split_offsets(length, n, out.ctypes.data)
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def pack(strings):
    """list[str|bytes] -> (offsets u32[n+1], values u8[...])"""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
    offs = np.zeros(len(bs) + 1, dtype=np.uint32)
    if bs:
        offs[1:] = np.cumsum([len(x) for x in bs], dtype=np.uint64).astype(np.uint32)
    vals = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return offs, vals


def batch(measure, offs_a, vals_a, offs_b, vals_b, nthreads=1):
    """Arrow-style columns (u32 or u64 offsets) -> f64[n]; broadcasting when one side has one row."""
    ra, rb = len(offs_a) - 1, len(offs_b) - 1
    n = max(ra, rb) if (ra == 1 or rb == 1) else ra
    out = np.empty(n, dtype=np.float64)
    va = np.ascontiguousarray(vals_a, dtype=np.uint8)
    vb = np.ascontiguousarray(vals_b, dtype=np.uint8)
    if offs_a.dtype == np.uint32:
        fn = lib().oracle_batch_u32
        oa = np.ascontiguousarray(offs_a, dtype=np.uint32)
        ob = np.ascontiguousarray(offs_b, dtype=np.uint32)
    else:
        fn = lib().oracle_batch
        oa = np.ascontiguousarray(offs_a, dtype=np.uint64)
        ob = np.ascontiguousarray(offs_b, dtype=np.uint64)
    # keep 1 byte so .ctypes.data is valid for empty value buffers
    if va.size == 0:
        va = np.zeros(1, dtype=np.uint8)
    if vb.size == 0:
        vb = np.zeros(1, dtype=np.uint8)
    rc = fn(MEASURE_ID[measure], oa.ctypes.data, va.ctypes.data, ra, ob.ctypes.data, vb.ctypes.data, rb,
            out.ctypes.data, nthreads)
    if rc != 0:
        raise ValueError("Inputs must have the same length, or one of them must be a Utf8 literal.")
    return out


def batch_strings(measure, A, B, nthreads=1):
    oa, va = pack(A)
    ob, vb = pack(B)
    return batch(measure, oa, va, ob, vb, nthreads)
