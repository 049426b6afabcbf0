"""Row sharding over ranks + the one collective of the path: gathering the f64 result column.

Rows are independent (every compute(a_i, b_i) is a pure function; the reference already splits rows
over threads, strsim.rs:73-97), so a column is cut into contiguous row ranges by the reference's own
`split_offsets` rule (strsim.rs:21-39: len/n each, remainder to the last rank) and each rank runs the
kernels on its shard.  No data-path collective is needed for the compute; the only exchange is the
final gather of the result shards to rank 0 (RCCL over xGMI when the backend is "nccl", gloo on CPU in
tests).  `torch.distributed.gather` needs equal counts, so shards are padded to the largest one.
"""
import torch
import torch.distributed as dist

from .context import split_offsets


def shard_range(n_rows, world_size, rank):
    """(offset, len) of `rank`'s rows -- reference split_offsets(len, n)[rank]."""
    return split_offsets(n_rows, world_size)[rank]


def gather_column(local, n_rows, dst=0, group=None, async_op=False, recv_buffer=None):
    """Gather the per-rank result shards (1-D tensors laid out by shard_range) onto `dst`.

    Returns the assembled [n_rows] tensor on dst (None elsewhere); with async_op=True returns
    (work, finish) where finish() yields that tensor after work.wait().
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    parts = split_offsets(n_rows, world)
    cap = max(p[1] for p in parts)
    if local.numel() != parts[rank][1]:
        raise ValueError(f"rank {rank}: shard has {local.numel()} rows, expected {parts[rank][1]}")
    send = local
    if local.numel() != cap:
        send = torch.zeros(cap, dtype=local.dtype, device=local.device)
        send[: local.numel()] = local
    bufs = None
    if rank == dst:
        if recv_buffer is None:
            recv_buffer = torch.empty(world * cap, dtype=local.dtype, device=local.device)
        bufs = [recv_buffer[i * cap:(i + 1) * cap] for i in range(world)]
    work = dist.gather(send, bufs, dst=dst, group=group, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        if all(p[1] == cap for p in parts):
            return recv_buffer[: n_rows]
        return torch.cat([bufs[i][: parts[i][1]] for i in range(world)])

    if async_op:
        return work, finish
    return finish()
