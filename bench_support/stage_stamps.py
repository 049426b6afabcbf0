"""Diagnostic: per-phase cycle sums of k_lane_stage (library built with EXTRA="-DSTRSIM_LAB -DSTRSIM_STAGE_STAMPS", selected by STRSIM_AMD_LIB)."""
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
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
dev = torch.device("cuda", 0)
measure, _, law, lo, hi, seed = W.CONFIGS[cfg]
if len(sys.argv) > 3:
    measure = sys.argv[3]
oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, 0, rows, dev)
out = torch.empty(rows, dtype=torch.float64, device=dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream)
for _ in range(30):
    ctx.pairs_device(measure, oa, va, ob, vb, out=out)
ctx.synchronize()
torch.cuda.synchronize()
L = S.lib()
f = L.strsim_debug_stage_stamps
f.argtypes = [C.c_void_p, C.c_size_t]
f.restype = C.c_int
wg = int(os.environ.get("STRSIM_STAGE_WG_PER_CU", "4")) * 256
nw = min(wg * 4, 16384)
buf = np.zeros((nw, 16), dtype=np.uint64)
assert f(buf.ctypes.data, nw) == 0
b = buf.astype(np.float64)
names = ["cut + bytes DMA issue", "store", "sortA", "barrier 1", "sortB", "DMA wait + barrier 2", "offsets DMA issue",
         "rounds: windows", "rounds: cores", "barrier G"]
tot = b[:, 10].mean()
print("%s %s" % (cfg, measure))
print("waves %d  cycles per wave %.0f  realtime ticks %.0f  -> clock %.3f GHz, kernel %.1f us" %
      (nw, tot, b[:, 11].mean(), tot / b[:, 11].mean() * 0.1, b[:, 11].mean() / 100.0))
for k, nme in enumerate(names):
    print("  %-24s %10.0f  %5.1f %%   (min %.0f max %.0f)" % (nme, b[:, k].mean(), 100 * b[:, k].mean() / tot, b[:, k].min(), b[:, k].max()))
print("  sum of phases %.1f %%" % (100 * b[:, :10].sum(axis=1).mean() / tot))

t0 = b[:, 12].min()
st, en = (b[:, 12] - t0) / 100.0, (b[:, 13] - t0) / 100.0
print("wave start us: min %.1f p50 %.1f max %.1f | wave end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" %
      (st.min(), np.median(st), st.max(), en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
life = en - st
print("wave lifetime us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max()))
# per workgroup (4 waves each): end time by workgroup index modulo 8 (XCD) and by position
wgend = en.reshape(-1, 4).max(axis=1)
for x in range(8):
    print("  XCD-group %d: workgroups %d, end p50 %.1f max %.1f" % (x, len(wgend[x::8]), np.median(wgend[x::8]), wgend[x::8].max()))

np.save(os.path.join(ROOT, "gpurun_out", "stage_stamps_raw.npy"), buf)
