#!/usr/bin/env python3
"""Column x literal throughput (the reference's `b.len() == 1` broadcast, strsim.rs:61-66), device-resident."""
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

_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
offA, valA, offB, valB, _, _ = W.device_columns(seed, law, lo, hi, 0, n, dev)
lo_, lv_ = S.pack_strings(["phillipsburgh"])
lit_off = torch.from_numpy(lo_.view(np.int32)).to(dev)
lit_val = torch.from_numpy(np.concatenate([lv_, np.zeros(64, np.uint8)])).to(dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream, one_launch=True)
for m in S.MEASURES:
    for name, args in (("col,col", (offA, valA, offB, valB)), ("col,lit", (offA, valA, lit_off, lit_val)),
                       ("lit,col", (lit_off, lit_val, offB, valB))):
        out = ctx.pairs_device(m, *args)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.pairs_device(m, *args, out=out)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{m:14s} {name}: {dt*1e3:7.3f} ms  {n/dt/1e9:6.2f} G pairs/s")
