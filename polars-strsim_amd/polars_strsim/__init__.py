"""polars_strsim -- drop-in package directory for the MI355X build.

Same five expression functions, signatures and `is_elementwise=True` registration as the reference
(reference polars_strsim/__init__.py:8-69).  Polars resolves the plugin by scanning this directory for a
shared library, finds libpolars_strsim_amd.so (built here by ../Makefile) and calls its
`_polars_plugin_<name>` symbols (include/polars_plugin_abi.h), which run on the GPU.
"""
import ctypes as _C
import importlib.util as _ilu
import os as _os
import sys as _sys
from pathlib import Path

import polars as pl
from polars._typing import IntoExpr
from polars.plugins import register_plugin_function

from polars_strsim.utils import parse_into_expr

_PLUGIN_DIR = Path(__file__).parent


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process (the same rule as strsim_amd/_lib.py, for the route where POLARS loads the library).

    Polars dlopen()s libpolars_strsim_amd.so on the first plugin call; it then binds /opt/rocm's libamdhip64.  A torch wheel
    brings its own copy, and an `import torch` AFTER that maps the second runtime next to the first -- whichever initialises
    second finds "No HIP GPUs".  When torch is installed but not imported yet, its copy is mapped first (by SONAME both
    then resolve to it); when torch is already imported nothing needs doing; without torch the system runtime is used.
    Set STRSIM_KEEP_SYSTEM_HIP=1 to switch this off."""
    if "torch" in _sys.modules or _os.environ.get("STRSIM_KEEP_SYSTEM_HIP"):
        return
    try:
        spec = _ilu.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = _os.path.join(_os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if _os.path.exists(bundled):
        _C.CDLL(bundled, mode=_C.RTLD_GLOBAL)


_share_hip_runtime_with_torch()


def _similarity(function_name: str, expr: IntoExpr, other: IntoExpr) -> pl.Expr:
    return register_plugin_function(
        plugin_path=_PLUGIN_DIR,
        function_name=function_name,
        args=[parse_into_expr(expr, dtype=pl.Utf8), parse_into_expr(other, dtype=pl.Utf8)],
        is_elementwise=True,
    )


def levenshtein(expr: IntoExpr, other: IntoExpr) -> pl.Expr:
    """Normalised Levenshtein similarity of two string columns (or a column and a literal)."""
    return _similarity("levenshtein", expr, other)


def jaro(expr: IntoExpr, other: IntoExpr) -> pl.Expr:
    """Jaro similarity."""
    return _similarity("jaro", expr, other)


def jaro_winkler(expr: IntoExpr, other: IntoExpr) -> pl.Expr:
    """Jaro-Winkler similarity (prefix scale 0.1, prefix cap 4, boost threshold 0.7)."""
    return _similarity("jaro_winkler", expr, other)


def jaccard(expr: IntoExpr, other: IntoExpr) -> pl.Expr:
    """Jaccard similarity of the two character multisets."""
    return _similarity("jaccard", expr, other)


def sorensen_dice(expr: IntoExpr, other: IntoExpr) -> pl.Expr:
    """Sorensen-Dice similarity of the two character multisets."""
    return _similarity("sorensen_dice", expr, other)


__all__ = [
    "levenshtein",
    "jaro",
    "jaro_winkler",
    "jaccard",
    "sorensen_dice",
]
