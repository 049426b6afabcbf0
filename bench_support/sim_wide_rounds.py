"""Replay of k_lane_wide's sort and round dealing on cfg3's lengths (host generator, no GPU): where a Jaro round's vector
instructions go by the cores' per-column costs, and what a different zip pass / key would change.  Diagnostic for DESIGN 3.3."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_support import workload as W

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
measure, _, law, lo, hi, seed = W.CONFIGS["cfg3"]
la = np.empty(rows, dtype=np.uint32)
lb = np.empty(rows, dtype=np.uint32)
assert W.lib().synth_lengths_host(seed, law, lo, hi, 0, rows, la.ctypes.data, lb.ctypes.data) == 0
mx, mn = np.maximum(la, lb), np.minimum(la, lb)
wide = (mx > 32) & (mx <= 128) & (mn >= 1)
print("rows %d, wide %.3f" % (rows, wide.mean()))
SUPER = 16384
tot = dict(p1=0.0, p2b=0.0, p2m=0.0, p2best=0.0, planes=0.0, fixed=0.0, rounds=0, lanes=0, useful_p1=0.0)
hist = {}
for s0 in range(0, rows, SUPER):
    sl = slice(s0, s0 + SUPER)
    w = wide[sl]
    t, p = mn[sl][w].astype(np.int64), mx[sl][w].astype(np.int64)
    cls = (p - 1) >> 5
    key = (cls * 32 + ((t - 1) >> 2)) * 4 + (p - 1 - 32 * cls) // 8
    order = np.argsort(key, kind="stable")
    t, p, cls = t[order], p[order], cls[order]
    n = len(t)
    # rounds cut from the long end
    hi_ = n
    while hi_ > 0:
        first = max(0, hi_ - 64)
        tt, pp = t[first:hi_], p[first:hi_]
        Wd = int((pp.max() - 1) >> 5) + 1
        ng4 = (int(tt.max()) + 3) >> 2
        nb4 = (int(pp.max()) + 3) >> 2
        k4 = (int(tt.max()) + 3) & ~3  # m <= text length: upper bound of the wave's largest m
        p1 = 4 * ng4 * (11 + 10 * Wd)
        p2b = 28 * nb4
        p2m = k4 * (5 + 9 * Wd)
        tot["p1"] += p1; tot["p2b"] += p2b; tot["p2m"] += p2m; tot["p2best"] += min(p2b, p2m)
        tot["planes"] += 58 * Wd; tot["fixed"] += 150
        tot["rounds"] += 1; tot["lanes"] += hi_ - first
        tot["useful_p1"] += float((tt * (11 + 10 * (((pp - 1) >> 5) + 1))).sum()) / 64.0
        hist[Wd] = hist.get(Wd, 0) + 1
        hi_ = first
r = tot["rounds"]
print("rounds %d (%.1f rows each), by width %s" % (r, tot["lanes"] / r, hist))
for k in ("p1", "p2b", "p2m", "p2best", "planes", "fixed"):
    print("  %-8s %8.1f per round" % (k, tot[k] / r))
print("  pass 1 useful (each lane its own length and width) %.1f per round" % (tot["useful_p1"] / r))
print("  total now (zip over b) %.1f, with the cheaper zip per round %.1f" % ((tot["p1"] + tot["p2b"] + tot["planes"] + tot["fixed"]) / r,
      (tot["p1"] + tot["p2best"] + tot["planes"] + tot["fixed"]) / r))
