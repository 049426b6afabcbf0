#!/usr/bin/env python3
"""Levenshtein on long ASCII strings (cfg5 lengths): a-z (five-plane batches) vs mixed case (seven planes)."""
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
oa, va, ob, vb = W.host_columns(5, W.UNIFORM, 1, 1024, 0, n)


def mixed(val, seed):
    """Upper-case a deterministic half of the letters: bits 5 now vary inside every string."""
    r = np.random.default_rng(seed).integers(0, 2, len(val), dtype=np.uint8)
    return (val - (r << 5)).astype(np.uint8)


dev = torch.device("cuda", 0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream, one_launch=True)
t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
pad = np.zeros(64, dtype=np.uint8)
cells = float((np.diff(oa.astype(np.int64)) * np.diff(ob.astype(np.int64))).sum())
# the same upper-casing pattern on both sides keeps edited copies close
for label, (xa, xb) in (("a-z", (va, vb)), ("mixed case", (mixed(va, 1), mixed(vb, 2)))):
    args = (t(oa, np.int32), t(np.concatenate([xa, pad]), np.uint8), t(ob, np.int32), t(np.concatenate([xb, pad]), np.uint8))
    out = ctx.pairs_device("levenshtein", *args)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.pairs_device("levenshtein", *args, out=out)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{label:10s}: {dt*1e3:8.3f} ms  {n/dt/1e6:7.2f} M pairs/s  {cells/dt/1e12:6.2f} TCUPS  (wave-kernel rows: {ctx.last_wave_rows})")
