"""strsim_amd -- host side of the MI355X pairwise string-similarity path (above the C ABI).

`Context` wraps include/strsim_amd.h; the five functions below mirror the reference's operator surface
(polars_strsim/__init__.py:8-60: levenshtein / jaro / jaro_winkler / jaccard / sorensen_dice over two
string columns, either side may be a single literal) for plain Python / numpy inputs, so parity tests
read like the reference's own.  Nulls (None) propagate: null in -> null (NaN in the f64 array plus a
validity mask), as in README.md:69-70.
"""
import numpy as np

from ._lib import MEASURES, MEASURE_ID, LIB_PATH, STATUS, ShapeMismatch, StrsimError, lib
from .context import Codec, Context, device_count, pack_strings, split_offsets

_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _as_column(x):
    """-> (list of str/bytes with None replaced by '', validity bool array or None)"""
    if isinstance(x, (str, bytes)) or x is None:
        x = [x]
    x = list(x)
    valid = np.array([v is not None for v in x], dtype=bool)
    vals = [v if v is not None else "" for v in x]
    return vals, (None if valid.all() else valid)


def similarity(measure, a, b, ctx=None):
    """f64 numpy array (NaN where either input is null) for two columns / a column and a literal."""
    ctx = ctx or default_context()
    A, va = _as_column(a)
    B, vb = _as_column(b)
    ao, av = pack_strings(A)
    bo, bv = pack_strings(B)
    out = ctx.pairs_host(measure, ao, av, bo, bv)
    n = out.size
    for v in (va, vb):
        if v is not None:
            out = out.copy()
            out[~(np.broadcast_to(v, (n,)) if v.size == 1 else v)] = np.nan
    return out


def levenshtein(a, b, ctx=None):
    return similarity("levenshtein", a, b, ctx)


def jaro(a, b, ctx=None):
    return similarity("jaro", a, b, ctx)


def jaro_winkler(a, b, ctx=None):
    return similarity("jaro_winkler", a, b, ctx)


def jaccard(a, b, ctx=None):
    return similarity("jaccard", a, b, ctx)


def sorensen_dice(a, b, ctx=None):
    return similarity("sorensen_dice", a, b, ctx)


__all__ = ["Codec", "Context", "device_count", "pack_strings", "split_offsets", "similarity", "levenshtein", "jaro",
           "jaro_winkler", "jaccard", "sorensen_dice", "MEASURES", "MEASURE_ID", "STATUS", "ShapeMismatch", "StrsimError"]
