"""Argument coercion for the plugin wrappers (same contract as the reference's polars_strsim/utils.py:6-43)."""
from __future__ import annotations

import polars as pl
from polars._typing import IntoExpr, PolarsDataType


def parse_into_expr(
    expr: IntoExpr,
    *,
    str_as_lit: bool = False,
    list_as_lit: bool = True,
    dtype: PolarsDataType | None = None,
) -> pl.Expr:
    """Turn one wrapper argument into a `pl.Expr`.

    * a `pl.Expr` is passed through untouched;
    * a `str` names a column, unless `str_as_lit` asks for a string literal;
    * a `list` becomes a literal; with `list_as_lit=False` it becomes a Series literal instead;
    * anything else becomes `pl.lit(value, dtype=dtype)`.
    """
    if isinstance(expr, pl.Expr):
        return expr
    if isinstance(expr, str) and not str_as_lit:
        return pl.col(expr)
    if isinstance(expr, list) and not list_as_lit:
        return pl.lit(pl.Series(expr), dtype=dtype)
    return pl.lit(expr, dtype=dtype)
