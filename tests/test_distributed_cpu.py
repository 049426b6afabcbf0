"""N > 1 path on CPU: two gloo ranks shard a frame by the reference's split_offsets rule, compute their shards
(the oracle stands in for the GPU kernels -- this test is about sharding + gather, not kernels) and gather
the f64 column on rank 0, which must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, measure, async_op, q):
    for p in (os.path.join(ROOT, "polars-strsim_amd"), HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gen
        import oracle_lib as O
        from strsim_amd.distributed import gather_column, shard_range
        A, B = gen.pairs(77, n_rows, gen.ASCII_LOWER, 0, 20)
        off, ln = shard_range(n_rows, world, rank)
        local = torch.from_numpy(O.batch_strings(measure, A[off:off + ln], B[off:off + ln])) if ln else torch.empty(0, dtype=torch.float64)
        if async_op:
            work, finish = gather_column(local, n_rows, dst=0, async_op=True)
            work.wait()
            full = finish()
        else:
            full = gather_column(local, n_rows, dst=0)
        if rank == 0:
            exp = O.batch_strings(measure, A, B)
            q.put(bool((full.numpy().view(np.uint64) == exp.view(np.uint64)).all()) and full.numel() == n_rows)
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,async_op", [(1001, False), (64, True), (3, False)])
def test_two_rank_shard_and_gather(n_rows, async_op):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, "levenshtein", async_op, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_ranges_cover_rows():
    sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
    from strsim_amd.distributed import shard_range
    for n, w in [(100, 8), (7, 8), (0, 2), (12_500_001, 8)]:
        parts = [shard_range(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and sum(p[1] for p in parts) == n
        for a, b in zip(parts, parts[1:]):
            assert a[0] + a[1] == b[0] or a[1] == 0


def test_shard_ranges_reference_rule_and_the_root_share_deviation():
    """shard_ranges: split_offsets(rows, N) by default (strsim.rs:21-39: rows / N each, the remainder to the last rank); with
    root_share < 1 rank 0 holds that fraction of an equal share in whole 64-row chunks and the others split the rest by the same rule --
    contiguous, complete, in order."""
    for p in (os.path.join(ROOT, "polars-strsim_amd"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    from strsim_amd.distributed import shard_ranges
    import oracle_lib as O
    for n, w in ((100_000_000, 8), (1_000_001, 2), (7, 8), (0, 3), (12345, 1)):
        assert shard_ranges(n, w) == O.split_offsets(n, w)
    for n, w, share in ((100_000_000, 8, 0.4), (1_000_000, 2, 0.5), (1_000_001, 3, 0.0), (999, 4, 0.9)):
        parts = shard_ranges(n, w, share)
        assert len(parts) == w and parts[0][0] == 0 and parts[0][1] % 64 == 0 and parts[0][1] <= n // w
        assert abs(parts[0][1] - n // w * share) < 64
        at = 0
        for off, ln in parts:
            assert off == at
            at += ln
        assert at == n
        assert [ln for _o, ln in parts[1:]] == [ln for _o, ln in O.split_offsets(n - parts[0][1], w - 1)]
    with pytest.raises(ValueError):
        shard_ranges(100, 2, 1.5)
