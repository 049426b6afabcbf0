#!/usr/bin/env python3
"""The five measures on mid-length ASCII strings (U{33..128} bytes, a-z: the k_lane_wide path), device-resident.
usage: bench_mid_ascii.py [rows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, ROOT)
import torch

import strsim_amd as S
from bench_support import workload as W

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream, one_launch=True)
oa, va, ob, vb, _, _ = W.device_columns(33, W.UNIFORM, 33, 128, 0, rows, dev)
out = torch.empty(rows, dtype=torch.float64, device=dev)
for m in S.MEASURES:
    for _ in range(2):
        ctx.pairs_device(m, oa, va, ob, vb, out=out)
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.pairs_device(m, oa, va, ob, vb, out=out)
    ctx.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-14s %8.2f ms per %d rows = %7.1f M pairs/s  (checksum %.6f)" % (m, dt * 1e3, rows, rows / dt / 1e6, float(out.sum())))
