#!/usr/bin/env python3
"""Levenshtein on long NON-ASCII strings (Cyrillic, 1..512 letters = 2..1024 bytes): the 32-row block step of k_wave_pairs
over 16-bit symbols (SYMBOLS batches)."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
oa, va, ob, vb = W.host_columns(5, W.UNIFORM, 1, 512, 0, n)


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
cells = float((np.diff(oa.astype(np.int64)) * np.diff(ob.astype(np.int64))).sum())
(ca0, ca1), (cb0, cb1) = cyr(oa, va), cyr(ob, vb)
args = (t(ca0, np.int32), t(np.concatenate([ca1, pad]), np.uint8), t(cb0, np.int32), t(np.concatenate([cb1, pad]), np.uint8))
out = ctx.pairs_device("levenshtein", *args)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ctx.pairs_device("levenshtein", *args, out=out)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"cyrillic 1..512 letters: {dt*1e3:8.3f} ms  {n/dt/1e6:8.2f} M pairs/s  {cells/dt/1e12:6.2f} TCUPS  (wave-kernel rows: {ctx.last_wave_rows})")
