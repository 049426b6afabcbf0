"""The Python wrapper surface (needs polars; skipped where it is not installed -- it is not in this image)."""
import pytest

pl = pytest.importorskip("polars")


def test_wrapper_signatures_and_all():
    import polars_strsim as ps
    assert ps.__all__ == ["levenshtein", "jaro", "jaro_winkler", "jaccard", "sorensen_dice"]
    for name in ps.__all__:
        e = getattr(ps, name)("name_a", "name_b")
        assert isinstance(e, pl.Expr)


@pytest.mark.gpu
def test_readme_demo_frame():
    import polars_strsim as ps
    df = pl.DataFrame({"name_a": ["phillips", "phillips", "", "", None, None],
                       "name_b": ["phillips", "philips", "phillips", "", "phillips", None]}).with_columns(
        levenshtein=ps.levenshtein("name_a", "name_b"), jaro=ps.jaro("name_a", "name_b"),
        jaro_winkler=ps.jaro_winkler("name_a", "name_b"), jaccard=ps.jaccard("name_a", "name_b"),
        sorensen_dice=ps.sorensen_dice("name_a", "name_b"))
    assert df["levenshtein"].to_list() == [1.0, 0.875, 0.0, 1.0, None, None]
