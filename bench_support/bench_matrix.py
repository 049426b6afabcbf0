#!/usr/bin/env python3
"""Throughput map: every measure x length class x {ASCII, 2-byte UTF-8}.  Finds the slow corners; kernel-only
(device-resident columns, results left on the device)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import strsim_amd as S
from bench_support import workload as W

MEASURES = ("levenshtein", "jaro", "jaro_winkler", "jaccard", "sorensen_dice")
# (label, lo, hi, rows)
CLASSES = (("1-32", 1, 32, 4_000_000), ("33-128", 33, 128, 2_000_000), ("129-1024", 129, 1024, 200_000))


def cyr(off, val):
    v = val.astype(np.uint16) - ord("a") + 0x430
    out = np.empty(2 * len(val), dtype=np.uint8)
    out[0::2], out[1::2] = (0xC0 | (v >> 6)).astype(np.uint8), (0x80 | (v & 0x3F)).astype(np.uint8)
    return (off.astype(np.uint64) * 2).astype(np.uint32), out


dev = torch.device("cuda", 0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream, one_launch=True)
t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
pad = np.zeros(64, dtype=np.uint8)
print(f"{'chars':9s} {'script':8s} " + " ".join(f"{m:>14s}" for m in MEASURES) + "   (M pairs/s)")
for label, lo, hi, n in CLASSES:
    oa, va, ob, vb = W.host_columns(11, W.UNIFORM, lo, hi, 0, n)
    for script, (ca, cb) in (("ascii", ((oa, va), (ob, vb))), ("cyrillic", (cyr(oa, va), cyr(ob, vb)))):
        args = (t(ca[0], np.int32), t(np.concatenate([ca[1], pad]), np.uint8), t(cb[0], np.int32), t(np.concatenate([cb[1], pad]), np.uint8))
        cells = []
        for m in MEASURES:
            out = ctx.pairs_device(m, *args)
            ctx.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                ctx.pairs_device(m, *args, out=out)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
            cells.append(n / dt / 1e6)
        print(f"{label:9s} {script:8s} " + " ".join(f"{c:14.1f}" for c in cells))
