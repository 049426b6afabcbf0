"""Loaders for the committed golden fixtures (tests/golden/)."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def reference_vectors():
    """[(measure, test_fn, a, b, expected)] -- the reference's 1 115 known-answer triples."""
    rows = []
    with open(os.path.join(GOLDEN, "reference_vectors.tsv"), encoding="utf-8") as f:
        for ln in f:
            if ln.startswith("#"):
                continue
            parts = ln.rstrip("\n").split("\t")
            assert len(parts) == 5, parts
            rows.append((parts[0], parts[1], parts[2], parts[3], float(parts[4])))
    return rows


def readme_table():
    with open(os.path.join(GOLDEN, "readme_table.json"), encoding="utf-8") as f:
        return json.load(f)["rows"]
