"""Diagnostic: per-phase cycle sums of the producer / consumer form of k_lane_stage (lab library built with
EXTRA="-DSTRSIM_LAB -DSTRSIM_STAGE_PC=1 -DSTRSIM_STAGE_STAMPS", selected by STRSIM_AMD_LIB).  python bench_support/stage_pc_stamps.py [rows] [measure]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, ROOT)
import torch
import strsim_amd as S
from bench_support import workload as W

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
measure = sys.argv[2] if len(sys.argv) > 2 else "levenshtein"
wg_per_cu = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, 0, rows, dev)
out = torch.empty(rows, dtype=torch.float64, device=dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream)
for _ in range(30):
    ctx.pairs_device(measure, oa, va, ob, vb, out=out)
ctx.synchronize()
torch.cuda.synchronize()
f = S.lib().strsim_debug_stage_stamps
f.argtypes = [C.c_void_p, C.c_size_t]
f.restype = C.c_int
nw = 256 * wg_per_cu * 8
buf = np.zeros((nw, 16), dtype=np.uint64)
assert f(buf.ctypes.data, nw) == 0
b = buf.astype(np.float64)
prod = b[b[:, 14] == 1.0]
cons = b[b[:, 14] == 0.0]
print("cfg2 %s, %d rows: %d producer waves, %d consumer waves, epochs per workgroup %.0f" % (measure, rows, len(prod), len(cons), b[:, 15].mean()))
for name, w, cats in (("PRODUCER", prod, [(0, "store"), (1, "cut + bytes DMA issue + sortA"), (2, "wait at X1"), (3, "sortB + offsets DMA issue"), (4, "DMA wait"), (5, "wait at X2")]),
                      ("CONSUMER", cons, [(6, "descriptor + windows"), (7, "cores"), (8, "wait at X1"), (9, "wait at X2")])):
    tot = w[:, 10].mean()
    print("%s wave: cycles %.0f, realtime ticks %.0f -> clock %.3f GHz, life %.1f us" % (name, tot, w[:, 11].mean(), tot / w[:, 11].mean() * 0.1, w[:, 11].mean() / 100.0))
    for k, nme in cats:
        print("  %-32s %10.0f  %5.1f %%   (min %.0f max %.0f)" % (nme, w[:, k].mean(), 100 * w[:, k].mean() / tot, w[:, k].min(), w[:, k].max()))
    print("  sum of phases %.1f %%" % (100 * w[:, [k for k, _ in cats]].sum(axis=1).mean() / tot))
