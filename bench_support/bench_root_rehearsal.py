#!/usr/bin/env python3
"""Rank 0's load at N = 8, rehearsed on ONE GPU (no RCCL involved -- there is no node to run it on).

At strong scaling over 8 GPUs rank 0 does three things per step, on three streams:
  compute  the kernels over its own shard (cfg2: 12.5 M rows of the 100 M-row frame, split_offsets(100 M, 8), strsim.rs:21-39)
  comm     codes its shard (strsim_codec_encode_packed) and receives 7 peers' segments over xGMI into the gather buffer
  decode   strsim_codec_decode_gathered over all 8 segments -> the 100 M-row f64 column (112 MB read, 800 MB written)
Here the peers' segments are copies of rank 0's own coded shard, "received" by device-to-device copies on the comm stream (they
write the same bytes into HBM the links would; a link delivers 14 MB in ~0.1 ms, which this rehearsal does not model).  The
raw-f64 transport (north_star's literal form) needs no decode: the root only takes 7 x 100 MB of incoming writes per step,
stood in for by a device-to-device copy of that size.
Prints ms per step of every leg alone and of the overlapped pipeline; DESIGN.md section 7 derives the N = 8 projection from it.
usage: bench_root_rehearsal.py [world=8] [steps=40]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, ROOT)
import torch

import strsim_amd as S
from bench_support import workload as W

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
measure, total_rows, law, lo, hi, seed = W.CONFIGS["cfg2"]
parts = S.split_offsets(total_rows, world)
row0, rows = parts[0]
dev = torch.device("cuda", 0)
compute, comm, decode = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.set_stream(compute)
ctx = S.Context(0, stream=compute.cuda_stream)  # stream-ordered (the ABI default): what the gather needs
ctx_comm = S.Context(0, stream=comm.cuda_stream)
ctx_dec = S.Context(0, stream=decode.cuda_stream)
oa, va, ob, vb, _, _ = W.device_columns(seed, law, lo, hi, row0, rows, dev)
outs = [torch.empty(rows, dtype=torch.float64, device=dev) for _ in range(2)]
codec = S.Codec(ctx, measure, 32)
EXC = 65536
code_bytes = (8 * codec.packed_words(rows) + 15) & ~15
ship_bytes = code_bytes + 16 + 12 * EXC
mine = torch.zeros(ship_bytes, dtype=torch.uint8, device=dev)
gathered = [torch.zeros(world * ship_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
column = torch.empty(total_rows, dtype=torch.float64, device=dev)
raw_in = torch.empty((world - 1) * rows, dtype=torch.float64, device=dev)
raw_src = torch.empty((world - 1) * rows, dtype=torch.float64, device=dev)
overflow = torch.zeros(1, dtype=torch.int32, device=dev)


def exc_views(buf, base):
    return buf[base:base + 4].view(torch.int32), buf[base + 16:base + 16 + 4 * EXC].view(torch.int32), buf[base + 16 + 4 * EXC:base + 16 + 12 * EXC].view(torch.float64)


def kernel(i):
    ctx.pairs_device(measure, oa, va, ob, vb, out=outs[i & 1])


ROOT_PLAIN = True  # [r5] the root copies its own f64 shard into the column instead of coding and decoding it (distributed.py)


def ship_coded(i):  # comm stream: the own shard (coded, or copied in as it is), "receive" the peers' segments
    with torch.cuda.stream(comm):
        g = gathered[i & 1]
        if ROOT_PLAIN:
            column[:rows].copy_(outs[i & 1], non_blocking=True)
        else:
            codec.encode_packed(outs[i & 1], mine[:code_bytes].view(torch.int64), ctx=ctx_comm, exc=exc_views(mine, code_bytes))
        for r in range(1 if ROOT_PLAIN else 0, world):
            g[r * ship_bytes:(r + 1) * ship_bytes].copy_(mine, non_blocking=True)


def decode_coded(i):
    with torch.cuda.stream(decode):
        codec.decode_gathered(gathered[i & 1], ship_bytes, world, rows, parts[-1][1], True, code_bytes, EXC, column, overflow, ctx=ctx_dec,
                              first_seg=1 if ROOT_PLAIN else 0)


def ship_raw(i):  # comm stream: 7 peers' f64 shards arrive (stand-in: a copy of that size), the own shard is copied in
    with torch.cuda.stream(comm):
        raw_in.copy_(raw_src, non_blocking=True)
        column[:rows].copy_(outs[i & 1], non_blocking=True)


def timed(fn, n=steps, warm=8):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    ctx.synchronize(); ctx_comm.synchronize(); ctx_dec.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def pipeline(coded):
    done = [None, None]      # shipment of the step that last used outs[b] / gathered[b]
    decoded = [None, None]

    def step(i):
        b = i & 1
        if done[b] is not None:
            compute.wait_event(done[b])          # outs[b] may be overwritten once its shipment has read it
        kernel(i)
        ev = torch.cuda.Event(); ev.record(compute)
        comm.wait_event(ev)
        if coded:
            if decoded[b] is not None:
                comm.wait_event(decoded[b])      # gathered[b] is free once its decode has run
            ship_coded(i)
            e2 = torch.cuda.Event(); e2.record(comm)
            done[b] = e2
            decode.wait_event(e2)
            decode_coded(i)
            e3 = torch.cuda.Event(); e3.record(decode)
            decoded[b] = e3
        else:
            ship_raw(i)
            e2 = torch.cuda.Event(); e2.record(comm)
            done[b] = e2
    return step


print("root rehearsal: cfg2 %s, %d rows of %d on rank 0 of %d; %d-bit codes, %.1f MB per segment, %d segments per decode" %
      (measure, rows, total_rows, world, codec.bits, ship_bytes / 1e6, world))
# one coded shard to stand for every peer's
codec.encode_packed(outs[0], mine[:code_bytes].view(torch.int64), ctx=ctx, exc=exc_views(mine, code_bytes))
kernel(0)
codec.encode_packed(outs[0], mine[:code_bytes].view(torch.int64), ctx=ctx, exc=exc_views(mine, code_bytes))
ctx.synchronize()
k = timed(kernel)
print("  kernels alone (stream-ordered call: 5 launches)          %.3f ms per step" % k)
for plain in (False, True):
    ROOT_PLAIN = plain
    tag = "root's own shard copied in as f64" if plain else "root codes and decodes its own shard too (rounds 3-4)"
    e = timed(lambda i: ship_coded(i))
    print("  [%s]" % tag)
    print("    own shard + %d segment copies alone (comm stream)       %.3f ms per step" % (world - (1 if plain else 0), e))
    d = timed(lambda i: decode_coded(i))
    nd = world - (1 if plain else 0)
    print("    decode_gathered alone: %d x %.1f MB -> %d MB of f64      %.3f ms per step  (byte roofline %.3f ms at 8 TB/s)" %
          (nd, code_bytes / 1e6, nd * rows * 8 // 1000000, d, (nd * code_bytes + 8 * nd * rows) / 8e12 * 1e3))
    pc = timed(pipeline(True))
    print("    PIPELINE coded: kernels | own shard + receive | decode    %.3f ms per step" % pc)
r = timed(lambda i: ship_raw(i))
print("  raw f64: %d MB of incoming writes (as a copy) + own shard  %.3f ms per step" % ((world - 1) * rows * 8 // 1000000, r))
pr = timed(pipeline(False))
print("  PIPELINE raw f64: kernels | receive                        %.3f ms per step" % pr)
assert int(overflow.item()) == 0
# the decoded column's first segment must be the kernel's own output
assert torch.equal(column[:rows].view(torch.int64), outs[(steps - 1) & 1].view(torch.int64))
print("  (decoded segment 0 == the kernel's own f64 shard, bit for bit)")
