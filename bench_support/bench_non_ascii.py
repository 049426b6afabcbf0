#!/usr/bin/env python3
"""Throughput on short NON-ASCII strings (Cyrillic names, 2 bytes per char): k_lane_utf8 vs what one-wave-per-pair costs."""
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

# take the ASCII synthetic frame (a-z) and map every letter to a Cyrillic letter (2 bytes): lengths 1..16 chars
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
oa, va, ob, vb = W.host_columns(7, W.UNIFORM, 1, 16, 0, n)


def cyr(off, val):
    v = val.astype(np.uint16) - ord("a") + 0x430            # U+0430..  -> D0 B0.. / D1 80..
    b0 = (0xC0 | (v >> 6)).astype(np.uint8)
    b1 = (0x80 | (v & 0x3F)).astype(np.uint8)
    out = np.empty(2 * len(val), dtype=np.uint8)
    out[0::2], out[1::2] = b0, b1
    return (off.astype(np.uint64) * 2).astype(np.uint32), out


dev = torch.device("cuda", 0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream, one_launch=True)
t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
pad = np.zeros(64, dtype=np.uint8)
for label, (ca, cb) in (("ascii a-z", ((oa, va), (ob, vb))), ("cyrillic", (cyr(oa, va), cyr(ob, vb)))):
    args = (t(ca[0], np.int32), t(np.concatenate([ca[1], pad]), np.uint8), t(cb[0], np.int32), t(np.concatenate([cb[1], pad]), np.uint8))
    for m in ("levenshtein", "jaro_winkler", "jaccard"):
        out = ctx.pairs_device(m, *args)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.pairs_device(m, *args, out=out)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{label:10s} {m:13s}: {dt*1e3:8.3f} ms  {n/dt/1e9:7.3f} G pairs/s  (rows left to the wave kernel: {ctx.last_wave_rows})")
