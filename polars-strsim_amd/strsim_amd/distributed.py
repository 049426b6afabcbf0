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


def shard_ranges(n_rows, world_size, root_share=1.0):
    """[(offset, len)] * world_size.  root_share == 1: the reference's partition, split_offsets(n_rows, world_size) (strsim.rs:21-39).
    Otherwise a DEVIATION for a root that also decodes the gathered column: rank 0 holds root_share of an
    equal share, in whole 64-row chunks, and the other ranks split the rest by the reference's rule."""
    parts = split_offsets(n_rows, world_size)
    if world_size <= 1 or root_share == 1.0:
        return parts
    if not 0.0 <= root_share <= 1.0:
        raise ValueError("root_share is a fraction of an equal share: 0 .. 1")
    r0 = int(n_rows // world_size * root_share) // 64 * 64
    rest = split_offsets(n_rows - r0, world_size - 1)
    return [(0, r0)] + [(r0 + o, ln) for o, ln in rest]


def gather_column(local, n_rows, dst=0, group=None, async_op=False, recv_buffer=None, parts=None):
    """Gather the per-rank result shards (1-D tensors laid out by shard_range) onto `dst`.

    `parts`: the ranks' (offset, len) when they are not split_offsets(n_rows, world) -- shard_ranges(..., root_share); the receive
    buffer (world x the longest shard) is laid out rank by rank either way.
    Returns the assembled [n_rows] tensor on dst (None elsewhere); with async_op=True returns
    (work, finish) where finish() yields that tensor after work.wait().
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if parts is None:
        parts = split_offsets(n_rows, world)
    elif len(parts) != world or sum(p[1] for p in parts) != n_rows:
        raise ValueError(f"gather_column: parts must hold one (offset, len) per rank and cover {n_rows} rows")
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


class AbiGather:
    """The C ABI's own gather (include/strsim_amd.h: strsim_gather_*): RCCL point-to-point sends of the ragged f64 shards into the
    root, enqueued on the context's stream -- what a host that binds the header (and has no torch.distributed) uses.  The 128-byte
    unique id of rank 0 has to reach every rank by the host's own means (a file, MPI, a socket; `unique_id()` makes it)."""
    ID_BYTES = 128

    @staticmethod
    def unique_id():
        import ctypes as C
        from ._lib import check, lib
        buf = (C.c_uint8 * AbiGather.ID_BYTES)()
        check(lib().strsim_gather_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, uid, world_size, rank):
        import ctypes as C
        from ._lib import check, lib
        self.ctx, self.world, self.rank = ctx, int(world_size), int(rank)
        self._h = C.c_void_p()
        ub = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(bytes(uid))
        check(lib().strsim_gather_create(ctx._h, ub, self.world, self.rank, C.byref(self._h)))

    def gather(self, shard, column, total_rows=None, root=0, parts=None):
        """shard: this rank's f64 rows (torch CUDA tensor, split_offsets(total_rows, world)[rank] of them); column: the root's
        [total_rows] f64 tensor (None elsewhere).  Asynchronous on the context's stream.
        `parts` = [(offset, len)] * world: an explicit partition instead (strsim_gather_f64_ranges, ABI 1.6)."""
        import ctypes as C
        from ._lib import check, lib
        sp = shard.data_ptr() if shard is not None and shard.numel() else None
        cp = column.data_ptr() if column is not None else None
        if parts is None:
            check(lib().strsim_gather_f64(self._h, sp, cp, int(total_rows), int(root)))
            return
        if len(parts) != self.world:
            raise ValueError("AbiGather.gather: one (offset, len) per rank")
        flat = (C.c_uint64 * (2 * self.world))(*[int(v) for p in parts for v in p])
        check(lib().strsim_gather_f64_ranges(self._h, sp, cp, flat, int(root)))

    def comm_count(self):
        """The rank count of the RCCL communicator itself (ncclCommCount through strsim_gather_comm_count)."""
        import ctypes as C
        from ._lib import check, lib
        n = C.c_int(0)
        check(lib().strsim_gather_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def close(self):
        from ._lib import lib
        if getattr(self, "_h", None) and self._h.value:
            lib().strsim_gather_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardGatherer:
    """Ships every step's result shard(s) to rank 0 on a side stream, overlapped with the next step's kernels.

    Shards follow the reference's row partition (`parts` = split_offsets(total rows, world), strsim.rs:21-39), so the last
    one may be longer; every rank ships the same number of bytes (the longest shard's), the root keeps each rank's own
    length.  Transport is the raw f64 column, or -- when a Codec covers the measure (strings <= `codec_chars`
    characters) -- its lossless code column, which rank 0 decodes back to f64 on a second side stream: 4x..7x fewer
    bytes on the one xGMI link each peer has into the root.  Values outside the codec's table (rows with longer strings)
    travel in an EXCEPTION BLOCK behind the codes -- count, rows, values, `exc_ship` entries -- and are patched in on the
    root with the count read on the device; a rank with more exceptions than the block holds makes `drain()` raise
    (nothing is ever left stale silently).  backend "nccl" (= RCCL) gathers device tensors; any other backend stages
    through the host (only meant to smoke-test the control flow on one GPU).
    """

    def __init__(self, ctx, compute_stream, measures, rows, device, backend="nccl", codec_chars=None, parts=None,
                 exc_ship=65536, root_plain=True, impl="torch"):
        """`compute_stream`: the torch stream whose handle `ctx` was created with (the kernels' stream).
        `rows`: this rank's shard length when `parts` is None (every rank the same), else ignored.
        `impl`: "torch" = torch.distributed.gather (equal counts: shards padded to the longest); "abi" = the C ABI's own gather
        (strsim_gather_f64_ranges: grouped RCCL send / recv of the ragged f64 shards straight into the root's column, the unique id
        broadcast through torch.distributed) -- raw f64 only, so `codec_chars` must be None.  Collective: every rank constructs."""
        import strsim_amd as S
        if ctx.stream != compute_stream.cuda_stream:
            raise ValueError("ShardGatherer: ctx does not enqueue on compute_stream")
        # shipments read a step's results behind an event on the compute stream, without retiring the call first: every kernel of
        # a call must be on the stream when submit() is called (strsim_ctx_set_stream_ordered; one-launch calls would finish slow
        # rows only at synchronize()).  The context STAYS in stream-ordered mode; calls enqueued before this point in one-launch
        # mode are retired here, so that nothing submitted later can be waiting for a late pass.
        ctx.set_stream_ordered(True)
        ctx.synchronize()
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.parts = list(parts) if parts is not None else [(r * rows, rows) for r in range(self.world)]
        if len(self.parts) != self.world:
            raise ValueError("ShardGatherer: one (offset, len) per rank")
        self.rows = self.parts[self.rank][1]
        self.cap_rows = max(p[1] for p in self.parts)
        self.total = sum(p[1] for p in self.parts)
        self.dev, self.host = device, backend != "nccl"
        self.compute = compute_stream
        self.comm = torch.cuda.Stream()    # encode + gather
        self.decode = torch.cuda.Stream()  # root only: decode of step i overlaps the gather of step i+1
        self.ctx = ctx
        self.ctx_comm = S.Context(ctx.device, stream=self.comm.cuda_stream)
        self.ctx_decode = S.Context(ctx.device, stream=self.decode.cuda_stream)
        self.nsub = 0
        self.root_plain = bool(root_plain)  # the root's own shard goes into the gathered column as it is (no encode / decode)
        self.timing = False      # timing_begin(): event pairs around every shipment (encode + gather) and every root decode
        self._t_ship, self._t_dec = [], []
        self.codecs = {}
        if codec_chars:
            for m in measures:
                try:
                    self.codecs[m] = S.Codec(ctx, m, codec_chars)
                except S.StrsimError:
                    self.codecs = {}
                    break
        self.codes, self.done = {}, {}
        self.impl, self.abi = impl, None
        if impl not in ("torch", "abi"):
            raise ValueError("ShardGatherer: impl is 'torch' or 'abi'")
        if impl == "abi":
            if self.codecs or codec_chars:
                raise ValueError("ShardGatherer: the C ABI's gather ships raw f64 (codec_chars=None)")
            box = [AbiGather.unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.abi = AbiGather(self.ctx_comm, box[0], self.world, self.rank)  # (collective: ncclCommInitRank)
        root = self.rank == 0
        self.recv = torch.empty(self.total, dtype=torch.float64, device=device) if root else None
        self.recv_codes = None
        self.decoded = [None, None]
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)  # root: exception blocks that did not hold everything
        # bytes one rank ships per column: packed codes (64 // bits per 64-bit word) when that beats 16 bits per row
        self.packed = bool(self.codecs) and all(64 // c.bits > 4 for c in self.codecs.values())
        if self.codecs:
            code_bytes = max(8 * c.packed_words(self.cap_rows) for c in self.codecs.values()) if self.packed else 2 * self.cap_rows
            self.code_bytes = (code_bytes + 15) & ~15
            self.exc_ship = int(exc_ship)
            self.ship_bytes = self.code_bytes + 16 + 12 * self.exc_ship  # codes | count (16 B) | rows u32 | values f64
        if self.codecs and root:  # codes travel as bytes: neither RCCL nor gloo has a 16-bit integer type
            self.recv_codes = [torch.empty(self.world * self.ship_bytes, dtype=torch.uint8, device="cpu" if self.host else device)
                               for _ in range(2)]
        elif root and self.host and self.abi is None:
            self.recv_host = torch.empty(self.world * self.cap_rows, dtype=torch.float64)
        if not self.codecs and root and not self.host and self.abi is None:
            self.recv_pad = torch.empty(self.world * self.cap_rows, dtype=torch.float64, device=device) \
                if any(p[1] != self.cap_rows for p in self.parts) else None

    def _exc_views(self, buf, base):
        """(count, rows, values) views of the exception block that starts `base` bytes into the uint8 tensor `buf`."""
        k = self.exc_ship
        cnt = buf[base:base + 4].view(torch.int32)
        rows = buf[base + 16:base + 16 + 4 * k].view(torch.int32)
        vals = buf[base + 16 + 4 * k:base + 16 + 12 * k].view(torch.float64)
        return cnt, rows, vals

    @property
    def transport(self):
        if not self.codecs:
            return "f64 (strsim_gather_f64_ranges: grouped RCCL send/recv)" if self.abi is not None else "f64"
        return "%d-bit codes, packed" % max(c.bits for c in self.codecs.values()) if self.packed else "u16 codes"

    def wait_slot(self, slot):
        """Call before overwriting the output buffer of `slot` (any hashable): its previous shipment must be done."""
        ev = self.done.get(slot)
        if ev is not None:
            self.compute.wait_event(ev)

    def submit(self, slot, measure, out):
        """`out`: this rank's f64 shard (parts[rank][1] rows) of `measure`, complete on the compute stream."""
        codec = self.codecs.get(measure)
        if out.numel() != self.rows:
            raise ValueError(f"rank {self.rank}: shard has {out.numel()} rows, expected {self.rows}")
        self.comm.wait_stream(self.compute)
        with torch.cuda.stream(self.comm):
            if self.timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(self.comm)
            if codec is not None:
                buf = self.codes.get(slot)
                if buf is None:
                    buf = self.codes[slot] = torch.zeros(self.ship_bytes, dtype=torch.uint8, device=self.dev)
                exc = self._exc_views(buf, self.code_bytes)
                # encode on the side stream too: it is a memory/latency-bound pass that overlaps the next step's kernels.
                # [r5] The ROOT does not code its own shard: it copies the f64 results into the gathered column (one device copy) and
                # ships a segment nobody reads -- every step waits for the root, which also decodes the peers' segments
                # (bench_support/bench_root_rehearsal.py: encode 0.09 ms + an eighth of the decode per step at N = 8).
                if self.rank == 0 and self.root_plain:
                    off0, ln0 = self.parts[0]
                    self.recv[off0:off0 + ln0].copy_(out, non_blocking=True)
                elif self.packed:
                    codec.encode_packed(out, buf[:self.code_bytes].view(torch.int64), ctx=self.ctx_comm, exc=exc)
                else:
                    codec.encode(out, buf[:2 * self.rows].view(torch.int16), ctx=self.ctx_comm, exc=exc)
                b = self.nsub & 1
                self.nsub += 1
                if self.rank == 0 and self.decoded[b] is not None:
                    self.comm.wait_event(self.decoded[b])  # the decode that last read this receive buffer
                work, _ = gather_column(buf.cpu() if self.host else buf, self.world * self.ship_bytes, dst=0, async_op=True,
                                        recv_buffer=self.recv_codes[b] if self.rank == 0 else None)
                work.wait()
                if self.rank == 0:
                    self.decode.wait_stream(self.comm)
                    with torch.cuda.stream(self.decode):
                        if self.timing:
                            d0 = torch.cuda.Event(enable_timing=True)
                            d0.record(self.decode)
                        rc = self.recv_codes[b].to(self.dev) if self.host else self.recv_codes[b]
                        chunk = self.parts[0][1]
                        if all(off == r * chunk for r, (off, _ln) in enumerate(self.parts)) and \
                                all(ln == chunk for _o, ln in self.parts[:-1]) and self.ship_bytes % 8 == 0:
                            # every rank coded its own shard; the shards are split_offsets' (equal, the last one longer): ONE launch
                            # decodes all segments and writes their exception blocks in
                            codec.decode_gathered(rc, self.ship_bytes, self.world, chunk, self.parts[-1][1], self.packed, self.code_bytes,
                                                  self.exc_ship, self.recv, self.overflow, ctx=self.ctx_decode,
                                                  first_seg=1 if self.root_plain else 0)
                        else:
                            for r, (off, ln) in enumerate(self.parts):  # any other partition: segment by segment
                                if r == 0 and self.root_plain:
                                    continue
                                seg = rc[r * self.ship_bytes:(r + 1) * self.ship_bytes]
                                dst = self.recv[off:off + ln]
                                if self.packed:
                                    codec.decode_packed(seg[:self.code_bytes].view(torch.int64), ln, dst, ctx=self.ctx_decode)
                                else:
                                    codec.decode(seg[:2 * ln].view(torch.int16), dst, ctx=self.ctx_decode)
                                codec.patch_indirect(self.recv, off, self._exc_views(seg, self.code_bytes), self.overflow,
                                                     ctx=self.ctx_decode)
                        self.decoded[b] = torch.cuda.Event(enable_timing=self.timing)
                        self.decoded[b].record(self.decode)
                        if self.timing:
                            self._t_dec.append((d0, self.decoded[b]))
            elif self.abi is not None:
                # ragged shards straight into their places of the root's column, on this (the comm) stream
                self.abi.gather(out, self.recv, parts=self.parts, root=0)
            else:
                ragged = any(p[1] != self.cap_rows for p in self.parts)
                src = out.cpu() if self.host else out
                if self.rank == 0:
                    rb = self.recv_host if self.host else (self.recv_pad if ragged else self.recv)
                else:
                    rb = None
                # (gather_column pads the shorter shards itself and needs the caller's row count to be the real total)
                work, _ = gather_column(src, self.total, dst=0, async_op=True, recv_buffer=rb, parts=self.parts)
                work.wait()
                if self.rank == 0 and ragged and not self.host:
                    for r, (off, ln) in enumerate(self.parts):
                        self.recv[off:off + ln].copy_(self.recv_pad[r * self.cap_rows:r * self.cap_rows + ln], non_blocking=True)
            ev = torch.cuda.Event(enable_timing=self.timing)
            ev.record(self.comm)
            self.done[slot] = ev
            if self.timing:
                self._t_ship.append((t0, ev))

    def result(self):
        """Rank 0, after drain(): the gathered f64 column of the last shipment (total rows); None elsewhere."""
        if self.rank != 0:
            return None
        if self.host and not self.codecs and self.abi is None:
            return torch.cat([self.recv_host[r * self.cap_rows:r * self.cap_rows + ln] for r, (_o, ln) in enumerate(self.parts)]).to(self.dev)
        return self.recv

    def comm_ranks(self):
        """impl "abi": the rank count RCCL's communicator reports for itself; None for torch.distributed's gather."""
        return self.abi.comm_count() if self.abi is not None else None

    def close(self):
        if self.abi is not None:
            self.abi.close()
            self.abi = None

    def drain(self):
        self.comm.synchronize()
        self.ctx_comm.synchronize()
        self.decode.synchronize()
        self.ctx_decode.synchronize()
        if self.rank == 0 and int(self.overflow.item()) != 0:
            raise RuntimeError("ShardGatherer: a rank had more rows outside the codec's table than its exception block "
                               f"holds ({self.exc_ship}); ship this column as f64 (codec_chars=None) or raise exc_ship")

    def timing_begin(self):
        """From now on every shipment and every root-side decode is bracketed by timed events on its own stream."""
        self.timing = True
        self._t_ship, self._t_dec = [], []

    def timing_end(self):
        """After drain(): (mean ms of a shipment on the comm stream: encode + gather as this rank saw it, mean ms of a
        root-side decode + exception patch, None on other ranks); resets."""
        ship = [a.elapsed_time(b) for a, b in self._t_ship]
        dec = [a.elapsed_time(b) for a, b in self._t_dec]
        self.timing = False
        self._t_ship, self._t_dec = [], []
        return (sum(ship) / len(ship) if ship else None), (sum(dec) / len(dec) if dec else None)

    def exceptions(self):
        """Rows outside the codec's table in the LAST shard coded (they travelled in the exception block): this rank's own, and on
        the root -- whose own shard is not coded when root_plain -- the largest count among the segments it received last."""
        n = 0
        for buf in self.codes.values():
            n = max(n, int(self._exc_views(buf, self.code_bytes)[0].item()))
        if self.rank == 0 and self.recv_codes is not None and self.nsub:
            rc = self.recv_codes[(self.nsub - 1) & 1]
            for r in range(1 if self.root_plain else 0, self.world):
                n = max(n, int(self._exc_views(rc, r * self.ship_bytes + self.code_bytes)[0].item()))
        return n
