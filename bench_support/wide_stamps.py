"""Diagnostic: per-phase cycle sums of k_lane_wide (library built with EXTRA="-DSTRSIM_LAB -DSTRSIM_WIDE_STAMPS", selected by STRSIM_AMD_LIB)."""
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

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
dev = torch.device("cuda", 0)
measure, _, law, lo, hi, seed = W.CONFIGS[cfg]
oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, 0, rows, dev)
out = torch.empty(rows, dtype=torch.float64, device=dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream)
for _ in range(4):
    ctx.pairs_device(measure, oa, va, ob, vb, out=out)
ctx.synchronize()
torch.cuda.synchronize()
L = S.lib()
f = L.strsim_debug_wide_stamps
f.argtypes = [C.c_void_p, C.c_size_t]
f.restype = C.c_int
nw = 16384
buf = np.zeros((nw, 16), dtype=np.uint64)
assert f(buf.ctypes.data, nw) == 0
b = buf.astype(np.float64)
b = b[b[:, 10] > 0]
names = ["mask + collect", "rows + offsets of a round", "windows, text to LDS, tests", "cores", "result + mask bit",
         "barrier behind the list"]
tot = b[:, 10].mean()
print("%s %s: waves %d  cycles per wave %.0f  realtime ticks %.0f -> clock %.3f GHz, wave lifetime %.1f us, rounds per wave %.1f" %
      (cfg, measure, len(b), tot, b[:, 11].mean(), tot / b[:, 11].mean() * 0.1, b[:, 11].mean() / 100.0, b[:, 6].mean()))
for k, nme in enumerate(names):
    print("  %-30s %10.0f  %5.1f %%   (min %.0f max %.0f)" % (nme, b[:, k].mean(), 100 * b[:, k].mean() / tot, b[:, k].min(), b[:, k].max()))
print("  sum of phases %.1f %%" % (100 * b[:, :6].sum(axis=1).mean() / tot))
