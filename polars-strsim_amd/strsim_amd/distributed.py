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


class ShardGatherer:
    """Ships every step's result shard(s) to rank 0 on a side stream, overlapped with the next step's kernels.

    Transport is the raw f64 column, or -- when a Codec covers the measure (strings <= `codec_chars` characters) --
    its lossless 16-bit code column, which rank 0 decodes back to f64 on the same side stream: 4x fewer bytes on the
    one xGMI link each peer has into the root.  backend "nccl" (= RCCL) gathers device tensors; any other backend
    stages through the host (only meant to smoke-test the control flow on one GPU).
    """

    def __init__(self, ctx, compute_stream, measures, rows, device, backend="nccl", codec_chars=None):
        """`compute_stream`: the torch stream whose handle `ctx` was created with (the kernels' stream)."""
        import strsim_amd as S
        if ctx.stream != compute_stream.cuda_stream:
            raise ValueError("ShardGatherer: ctx does not enqueue on compute_stream")
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.rows, self.dev, self.host = rows, device, backend != "nccl"
        self.compute = compute_stream
        self.comm = torch.cuda.Stream()    # encode + gather
        self.decode = torch.cuda.Stream()  # root only: decode of step i overlaps the gather of step i+1
        self.ctx = ctx
        self.ctx_comm = S.Context(ctx.device, stream=self.comm.cuda_stream)
        self.ctx_decode = S.Context(ctx.device, stream=self.decode.cuda_stream)
        self.nsub = 0
        self.codecs = {}
        if codec_chars:
            for m in measures:
                try:
                    self.codecs[m] = S.Codec(ctx, m, codec_chars)
                except S.StrsimError:
                    self.codecs = {}
                    break
        self.codes, self.done = {}, {}
        n = self.world * rows
        root = self.rank == 0
        self.recv = torch.empty(n, dtype=torch.float64, device=device) if root else None
        self.recv_codes = None
        self.decoded = [None, None]
        # bytes one rank ships per column: packed codes (64 // bits per 64-bit word) when that beats 16 bits per row
        self.packed = bool(self.codecs) and all(64 // c.bits > 4 for c in self.codecs.values())
        if self.codecs:
            self.ship_bytes = max(8 * c.packed_words(rows) for c in self.codecs.values()) if self.packed else 2 * rows
        if self.codecs and root:  # codes travel as bytes: neither RCCL nor gloo has a 16-bit integer type
            self.recv_codes = [torch.empty(self.world * self.ship_bytes, dtype=torch.uint8, device="cpu" if self.host else device)
                               for _ in range(2)]
        elif root and self.host:
            self.recv_host = torch.empty(n, dtype=torch.float64)

    @property
    def transport(self):
        if not self.codecs:
            return "f64"
        return "%d-bit codes, packed" % max(c.bits for c in self.codecs.values()) if self.packed else "u16 codes"

    def wait_slot(self, slot):
        """Call before overwriting the output buffer of `slot` (any hashable): its previous shipment must be done."""
        ev = self.done.get(slot)
        if ev is not None:
            self.compute.wait_event(ev)

    def submit(self, slot, measure, out):
        codec = self.codecs.get(measure)
        src = out
        self.comm.wait_stream(self.compute)
        with torch.cuda.stream(self.comm):
            if codec is not None:
                buf = self.codes.get(slot)
                if buf is None:
                    buf = self.codes[slot] = torch.empty(self.ship_bytes, dtype=torch.uint8, device=self.dev)
                # encode on the side stream too: it is a memory/latency-bound pass that overlaps the next step's
                # VALU-bound kernels
                if self.packed:
                    codec.encode_packed(out, buf.view(torch.int64), ctx=self.ctx_comm)
                else:
                    codec.encode(out, buf.view(torch.int16), ctx=self.ctx_comm)
                raw = buf
                b = self.nsub & 1
                self.nsub += 1
                if self.rank == 0 and self.decoded[b] is not None:
                    self.comm.wait_event(self.decoded[b])  # the decode that last read this receive buffer
                work, _ = gather_column(raw.cpu() if self.host else raw, self.world * self.ship_bytes, dst=0, async_op=True,
                                        recv_buffer=self.recv_codes[b] if self.rank == 0 else None)
                work.wait()
                if self.rank == 0:
                    self.decode.wait_stream(self.comm)
                    with torch.cuda.stream(self.decode):
                        rc = self.recv_codes[b].to(self.dev) if self.host else self.recv_codes[b]
                        if self.packed:  # every rank packed its own shard: decode them one by one
                            for r in range(self.world):
                                seg = rc[r * self.ship_bytes:(r + 1) * self.ship_bytes].view(torch.int64)
                                codec.decode_packed(seg, self.rows, self.recv[r * self.rows:(r + 1) * self.rows], ctx=self.ctx_decode)
                        else:
                            codec.decode(rc.view(torch.int16), self.recv, ctx=self.ctx_decode)
                        self.decoded[b] = torch.cuda.Event()
                        self.decoded[b].record(self.decode)
            else:
                work, _ = gather_column(src.cpu() if self.host else src, self.world * self.rows, dst=0, async_op=True,
                                        recv_buffer=(self.recv_host if self.host else self.recv) if self.rank == 0 else None)
                work.wait()
            ev = torch.cuda.Event()
            ev.record(self.comm)
            self.done[slot] = ev

    def drain(self):
        self.comm.synchronize()
        self.ctx_comm.synchronize()
        self.decode.synchronize()
        self.ctx_decode.synchronize()

    def exceptions(self):
        """Rows the codecs could not code in their LAST encode on this rank (must be 0 for the result to be complete)."""
        return sum(int(c.exc_count.item()) for c in self.codecs.values())
