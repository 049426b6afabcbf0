"""Child process of tests/test_gather_abi.py: ONE rank of a C-ABI gather (include/strsim_amd.h: strsim_gather_*).

  gather_child.py <world> <rank> <device> <id file> <rows> <root> <split|ranges:<root_share>>

Every rank computes Levenshtein on its shard of the same seeded frame with the product library, the shards are gathered onto `root`
on the context's stream (behind the kernels), and the root compares the whole column with the oracle bit for bit.  The 128-byte
unique id travels through a file (rank 0 writes it, the others wait for it): the host's "own means" of the header's contract.
Which collectives library carries the bytes is the parent's business (STRSIM_RCCL_LIB, or real RCCL with one GPU per rank).
Prints GATHER-OK <comm ranks> on success."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "polars-strsim_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch

import gen
import oracle_lib as O
import strsim_amd as S
from strsim_amd.distributed import AbiGather, shard_ranges

world, rank, device = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
id_file, rows, root, mode = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]

if rank == 0:
    uid = AbiGather.unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        assert time.time() - t0 < 120, "rank 0 never wrote the unique id"
        time.sleep(0.01)
    uid = open(id_file, "rb").read()
assert len(uid) == AbiGather.ID_BYTES

parts = S.split_offsets(rows, world) if mode == "split" else shard_ranges(rows, world, float(mode.split(":")[1]))
off, ln = parts[rank]
A, B = gen.pairs(1234, rows, gen.ASCII_LOWER, 0, 32)
dev = torch.device("cuda", device)
torch.cuda.set_device(dev)
with S.Context(device) as ctx:
    g = AbiGather(ctx, uid, world, rank)
    assert g.comm_count() == world
    for rep in range(2):  # (twice: the communicator is reused step after step)
        shard = None
        if ln:
            oa, va = S.pack_strings(A[off:off + ln])
            ob, vb = S.pack_strings(B[off:off + ln])
            t = lambda x, dt: torch.from_numpy(x.view(dt)).to(dev)
            pad = np.zeros(64, dtype=np.uint8)
            cols = (t(oa, np.int32), t(np.concatenate([va, pad]), np.uint8), t(ob, np.int32), t(np.concatenate([vb, pad]), np.uint8))
            torch.cuda.synchronize()
            shard = ctx.pairs_device("levenshtein", *cols)  # enqueued; the gather goes behind it on the same stream
        column = torch.full((rows,), -1.0, dtype=torch.float64, device=dev) if rank == root else None
        torch.cuda.synchronize()
        if mode == "split":
            g.gather(shard, column, total_rows=rows, root=root)
        else:
            g.gather(shard, column, root=root, parts=parts)
        ctx.synchronize()
        torch.cuda.synchronize()
        if rank == root:
            got = column.cpu().numpy()
            exp = O.batch_strings("levenshtein", A, B, 4)
            bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
            assert bad.size == 0, ("rows differ from the oracle", rep, bad[:5].tolist(), got[bad[:5]].tolist(), parts)
    g.close()
print("GATHER-OK", world, flush=True)
