"""polars_strsim -- drop-in package directory for the MI355X build.

Same five expression functions, signatures and `is_elementwise=True` registration as the reference
(reference polars_strsim/__init__.py:8-69).  Polars resolves the plugin by scanning this directory for a
shared library, finds libpolars_strsim_amd.so (built here by ../Makefile) and calls its
`_polars_plugin_<name>` symbols (include/polars_plugin_abi.h), which run on the GPU.
"""
from pathlib import Path

import polars as pl
from polars._typing import IntoExpr
from polars.plugins import register_plugin_function

from polars_strsim.utils import parse_into_expr

_PLUGIN_DIR = Path(__file__).parent


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
