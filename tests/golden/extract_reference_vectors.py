#!/usr/bin/env python3
"""Extract the reference's known-answer vectors into data fixtures.

Run in the build container (where /root/reference is mounted); the outputs are
committed, the reference itself never travels to the GPU box.

Sources (data only -- no reference code is copied):
  * /root/reference/src/expressions/strsim.rs:371-1534  -- 1 115 `x.test("a", "b", expected)`
    triples in ten #[test] functions (two per measure), checked there at abs tol 1e-8
    (strsim.rs:350).
  * /root/reference/README.md:59-72 -- the 6-row x 5-measure demo table (incl. null rows).

Outputs:
  tests/golden/reference_vectors.tsv   measure \t test_fn \t a \t b \t expected
  tests/golden/readme_table.json       rows with a/b (null-able) and the five printed values
"""
import json
import os
import re
import sys

REF = os.environ.get("STRSIM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

FN_TO_MEASURE = {
    "levenshtein": "levenshtein",
    "jaro": "jaro",
    "jaro_winkler": "jaro_winkler",
    "jaccard": "jaccard",
    "sorensen_dice": "sorensen_dice",
}


def extract_vectors():
    src = open(os.path.join(REF, "src/expressions/strsim.rs"), encoding="utf-8").read().splitlines()
    fn_re = re.compile(r"^\s*fn (\w+)_(edge_cases|test_cases)\(\)")
    t_re = re.compile(r'^\s*\w+\.test\("([^"\\]*)", "([^"\\]*)", ([0-9.eE+-]+)\);\s*$')
    rows, cur = [], None
    for ln in src:
        m = fn_re.match(ln)
        if m:
            cur = (FN_TO_MEASURE[m.group(1)], m.group(1) + "_" + m.group(2))
            continue
        m = t_re.match(ln)
        if m:
            assert cur is not None
            rows.append((cur[0], cur[1], m.group(1), m.group(2), m.group(3)))
        elif ".test(" in ln and "fn test" not in ln:
            raise SystemExit("unparsed vector line: " + ln)
    return rows


def extract_readme():
    lines = open(os.path.join(REF, "README.md"), encoding="utf-8").read().splitlines()
    hdr = None
    out = []
    for ln in lines:
        if not ln.startswith("|"):
            continue
        cells = [c.strip() for c in ln.strip().strip("|").split("|")]
        if cells[0] == "name_a":
            hdr = cells
            continue
        if hdr is None or cells[0] in ("---", "str"):
            continue
        if len(cells) != len(hdr):
            continue
        rec = {}
        for k, v in zip(hdr, cells):
            if k in ("name_a", "name_b"):
                rec[k] = None if v == "null" else v
            else:
                rec[k] = None if v == "null" else float(v)
        out.append(rec)
    return out


def main():
    rows = extract_vectors()
    with open(os.path.join(HERE, "reference_vectors.tsv"), "w", encoding="utf-8") as f:
        f.write("# measure\ttest_fn\ta\tb\texpected   (from reference src/expressions/strsim.rs:371-1534; tol 1e-8 per :350)\n")
        for r in rows:
            f.write("\t".join(r) + "\n")
    tab = extract_readme()
    with open(os.path.join(HERE, "readme_table.json"), "w", encoding="utf-8") as f:
        json.dump({"source": "reference README.md:59-72 / demo.py:4-15", "rows": tab}, f, indent=1)
    from collections import Counter
    print(len(rows), dict(Counter(r[0] for r in rows)), len(tab), file=sys.stderr)


if __name__ == "__main__":
    main()
