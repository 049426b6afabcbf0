import sys, time
sys.path.insert(0, "polars-strsim_amd"); sys.path.insert(0, ".")
import torch, strsim_amd as S
from bench_support import workload as W
_, _, law, lo, hi, seed = W.CONFIGS["cfg2"]
dev = torch.device("cuda", 0)
n = 100_000_000
offA, valA, offB, valB, _, _ = W.device_columns(seed, law, lo, hi, 0, n, dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = S.Context(0, stream=st.cuda_stream)
for m in ("levenshtein", "jaccard", "jaro_winkler"):
    out = ctx.pairs_device(m, offA, valA, offB, valB); ctx.synchronize()
    c = S.Codec(ctx, m, 32)
    codes = c.encode(out); dec = c.decode(codes); ctx.synchronize(); torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); 
    for _ in range(5): c.encode(out, codes)
    e1.record()
    for _ in range(5): c.decode(codes, dec)
    e2.record(); torch.cuda.synchronize()
    print(m, "entries", c.entries, "encode ms", e0.elapsed_time(e1)/5, "decode ms", e1.elapsed_time(e2)/5, "exc", int(c.exc_count.item()), "equal", torch.equal(dec.view(torch.int64), out.view(torch.int64)))
    words = c.encode_packed(out); dec2 = c.decode_packed(words, n); ctx.synchronize(); torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(5): c.encode_packed(out, words)
    e1.record()
    for _ in range(5): c.decode_packed(words, n, dec2)
    e2.record(); torch.cuda.synchronize()
    print(m, "packed: bits", c.bits, "bytes/row", 8 * words.numel() / n, "encode ms", e0.elapsed_time(e1)/5, "decode ms", e1.elapsed_time(e2)/5, "equal", torch.equal(dec2.view(torch.int64), out.view(torch.int64)))
