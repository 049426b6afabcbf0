"""Debug: launches per call and late rows on the cfg1 / cfg2 frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd")); sys.path.insert(0, ROOT)
import torch
import strsim_amd as S
from bench_support import workload as W
dev = torch.device("cuda", 0)
for cfg in ("cfg1", "cfg2"):
    m, rows, law, lo, hi, seed = W.CONFIGS[cfg]
    rows = min(rows, int(os.environ.get("MAXROWS", "4000000")))
    oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, 0, rows, dev)
    torch.cuda.synchronize()
    with S.Context(0) as ctx:
        for timing in (False, True):
            ctx.timing(timing)
            for i in range(4):
                b = ctx.enqueued_ops
                out = ctx.pairs_device(m, oa, va, ob, vb)
                ops = ctx.enqueued_ops - b
                ctx.synchronize()
                print(cfg, "timing", timing, "call", i, "ops", ops, "late", ctx.last_late_rows, "wave_rows", ctx.last_wave_rows)
