"""Simulation (CPU, no GPU): how many DP columns k_lane_stage<levenshtein> would run on cfg2 if a pair's COMMON PREFIX AND SUFFIX were stripped
before the block's rows are sorted into rounds (the distance does not change; half of cfg2's pairs are edited copies).  The kernel's own
dealing: blocks of 512 consecutive rows, rows sorted by text length (the shorter string) in buckets of two columns, rounds of 64 from the
long end, a round runs its longest text rounded up to a bucket.
  python bench_support/sim_stage_strip.py [rows]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import workload as W

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 400
_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, rows)
la = np.diff(oa).astype(np.int64)
lb = np.diff(ob).astype(np.int64)
pre = np.zeros(rows, dtype=np.int64)
suf = np.zeros(rows, dtype=np.int64)
for i in range(rows):
    a = va[oa[i]:oa[i + 1]]
    b = vb[ob[i]:ob[i + 1]]
    m = min(len(a), len(b))
    if m:
        d = np.nonzero(a[:m] != b[:m])[0]
        p = int(d[0]) if d.size else m
        a2, b2 = a[p:], b[p:]
        m2 = min(len(a2), len(b2))
        if m2:
            d2 = np.nonzero(a2[::-1][:m2] != b2[::-1][:m2])[0]
            s = int(d2[0]) if d2.size else m2
        else:
            s = 0
        pre[i], suf[i] = p, s


def columns_run(text):
    """mean columns a pair runs under the kernel's dealing; text = the text length per row (0: the row needs no column)"""
    run = 0
    for b0 in range(0, rows, 512):
        t = np.sort(text[b0:b0 + 512])[::-1]
        for r0 in range(0, len(t), 64):
            longest = t[r0]
            run += (((max(int(longest), 1) - 1) | 1) + 1) * min(64, len(t) - r0)
    return run / rows


text = np.minimum(la, lb)
print("cfg2, %d rows: mean text (shorter string) %.2f columns; the kernel's dealing runs %.2f per pair" % (rows, text.mean(), columns_run(text)))
tp = np.minimum(la - pre, lb - pre)
print("prefix stripped:          mean text %.2f, runs %.2f per pair (mean prefix %.2f)" % (tp.mean(), columns_run(tp), pre.mean()))
ts = np.minimum(la - pre - suf, lb - pre - suf)
print("prefix + suffix stripped: mean text %.2f, runs %.2f per pair (mean suffix behind the prefix %.2f)" % (ts.mean(), columns_run(ts), suf.mean()))
