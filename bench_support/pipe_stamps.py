"""Diagnostic: per-phase cycle sums of k_lane_pipe (library built with -DSTRSIM_PIPE_STAMPS, selected by STRSIM_AMD_LIB)."""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda", 0)
_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, 0, rows, dev)
out = torch.empty(rows, dtype=torch.float64, device=dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream)
for _ in range(30):
    ctx.pairs_device("levenshtein", oa, va, ob, vb, out=out)
ctx.synchronize()
torch.cuda.synchronize()
L = S.lib()
f = L.strsim_debug_pipe_stamps
f.argtypes = [C.c_void_p, C.c_size_t]
f.restype = C.c_int
wg = int(os.environ.get("STRSIM_STAMP_WG_PER_CU", "5")) * 256
nw = wg * 4
buf = np.zeros((nw, 10), dtype=np.uint64)
assert f(buf.ctypes.data, nw) == 0
b = buf.astype(np.float64)
names = ["store+sortB", "barrier Y", "offset loads", "window issue", "cores", "wait+moves", "sortA", "barrier X"]
tot = b[:, 8].mean()
print("waves %d  cycles per wave %.0f  realtime ticks %.0f  -> clock %.3f GHz, kernel %.1f us" %
      (nw, tot, b[:, 9].mean(), tot / b[:, 9].mean() * 0.1, b[:, 9].mean() / 100.0))
for k, nme in enumerate(names):
    print("  %-14s %10.0f  %5.1f %%   (min %.0f max %.0f)" % (nme, b[:, k].mean(), 100 * b[:, k].mean() / tot, b[:, k].min(), b[:, k].max()))
print("  sum of phases %.1f %%" % (100 * b[:, :8].sum(axis=1).mean() / tot))
