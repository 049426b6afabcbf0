"""Device context + the two compute entry points of the C ABI, for numpy (host) and torch (device) buffers."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, measure_id


def split_offsets(length, n):
    """Row partition of the reference's split_offsets (strsim.rs:21-39) -> [(offset, len)] * n."""
    out = np.zeros(2 * n, dtype=np.uint64)
    lib().strsim_split_offsets(int(length), int(n), out.ctypes.data)
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def device_count():
    return int(lib().strsim_device_count())


class Context:
    """One GPU + one HIP stream + workspace (include/strsim_amd.h: strsim_ctx_t).  One per thread."""

    def __init__(self, device=0, stream=None, one_launch=False):
        """`one_launch`: opt in to one-launch calls (strsim_ctx_set_stream_ordered(ctx, 0), ABI 1.4): a call that is expected to
        need the first kernel only is enqueued as that kernel alone, and slow rows it turns out to hold are finished when the
        call is retired -- for callers that synchronize() / retire_oldest() before they read results.  Default: every row of
        strings <= 1024 bytes is complete in stream order.
        `stream`: an int hipStream_t (e.g. torch.cuda.Stream().cuda_stream), or None for an own non-blocking stream.
        Handle 0 -- torch's DEFAULT stream, `torch.cuda.current_stream().cuda_stream` outside a `torch.cuda.stream(s)`
        block -- is refused: the C ABI reads NULL as "create your own stream", so the context would silently run on a
        stream that torch-side events and copies are not ordered against.  Pass a real torch stream and do the torch-side
        work (event.record(s), .cpu()) under `torch.cuda.stream(s)`, or pass None and use ctx.synchronize()."""
        if stream is not None and int(stream) == 0:
            raise ValueError("stream handle 0 is the default stream: pass None for an own stream, or a torch.cuda.Stream()'s "
                             "cuda_stream (and run the torch-side work under torch.cuda.stream(s))")
        self._h = C.c_void_p()
        check(lib().strsim_ctx_create(int(device), C.c_void_p(int(stream)) if stream is not None else None, C.byref(self._h)))
        self.device = int(device)
        if one_launch:
            self.set_stream_ordered(False)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().strsim_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def stream(self):
        return lib().strsim_ctx_stream(self._h)

    def synchronize(self):
        check(lib().strsim_ctx_synchronize(self._h))

    def timing(self, enable=True):
        check(lib().strsim_ctx_timing_enable(self._h, 1 if enable else 0))

    def timing_read(self):
        """-> dict(lane_ms, lane_launches, wave_ms, wave_launches) accumulated since the last read."""
        a, b = C.c_double(), C.c_double()
        na, nb = C.c_uint64(), C.c_uint64()
        check(lib().strsim_ctx_timing_read(self._h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return {"lane_ms": a.value, "lane_launches": na.value, "wave_ms": b.value, "wave_launches": nb.value}

    @property
    def last_wave_rows(self):
        return int(lib().strsim_ctx_last_wave_rows(self._h))

    @property
    def last_late_rows(self):
        """Rows finished by a pass that the last synchronize() / retire_oldest() launched (slow rows of a one-launch call,
        long strings): written AFTER whatever was enqueued on the stream behind the call."""
        return int(lib().strsim_ctx_last_late_rows(self._h))

    @property
    def enqueued_ops(self):
        """Kernels + device copies enqueued for pair calls so far (1 per call when no slow rows are expected, else 5)."""
        return int(lib().strsim_ctx_enqueued_ops(self._h))

    def set_stream_ordered(self, enable=True):
        """True (the default of a new context): every call enqueues all its kernels up front, so results (strings <= 1024
        bytes) are complete in stream order -- for callers that consume them behind an event without retiring the call.
        False: one-launch calls (see __init__).  Takes effect with the next call (strsim_ctx_set_stream_ordered)."""
        check(lib().strsim_ctx_set_stream_ordered(self._h, 1 if enable else 0))

    @property
    def stream_ordered(self):
        """The mode the next call is enqueued in (strsim_ctx_get_stream_ordered, ABI 1.6): True = every kernel up front."""
        return bool(lib().strsim_ctx_get_stream_ordered(self._h))

    @property
    def last_long_rows(self):
        """Rows with a string beyond the wave-kernel cap among the calls the last synchronize() retired."""
        return int(lib().strsim_ctx_last_long_rows(self._h))

    # ---- device-resident (torch tensors on this context's GPU) -------------------------------------
    def pairs_device(self, measure, a_offsets, a_values, b_offsets, b_values, out=None):
        """Enqueue one pass; tensors are torch CUDA tensors (offsets int32/uint32 [rows+1], values uint8).
        Returns the f64 output tensor; complete after synchronize()."""
        import torch
        ra, rb = a_offsets.numel() - 1, b_offsets.numel() - 1
        n = rb if ra == 1 else ra
        for t in (a_offsets, b_offsets):
            assert t.is_cuda and t.element_size() == 4 and t.is_contiguous()
        for t in (a_values, b_values):
            assert t.is_cuda and t.element_size() == 1 and t.is_contiguous()
        if out is None:
            out = torch.empty(n if (ra == rb or ra == 1 or rb == 1) else 0, dtype=torch.float64, device=a_offsets.device)
        check(lib().strsim_pairs_device(self._h, measure_id(measure),
                                        a_offsets.data_ptr(), a_values.data_ptr(), ra,
                                        b_offsets.data_ptr(), b_values.data_ptr(), rb,
                                        out.data_ptr(), out.numel()))
        return out

    def offsets_from_lengths(self, lengths, out=None):
        """uint8 device tensor of string lengths -> uint32 offsets (rows + 1) on the device (strsim_offsets_from_lengths)."""
        import torch
        n = lengths.numel()
        if out is None:
            out = torch.empty(n + 1, dtype=torch.int32, device=lengths.device)
        check(lib().strsim_offsets_from_lengths(self._h, lengths.data_ptr(), int(n), out.data_ptr()))
        return out

    def column_from_views(self, views, long_values, packed_bytes, bounded=True):
        """Utf8View slots on the device (uint8 tensor of rows x 16 bytes; slots of strings beyond 12 bytes carry their offset in
        `long_values` in their last word) -> (offsets int32 [rows + 1], values uint8 [packed_bytes + 64]).  `bounded` (ABI 1.5,
        strsim_column_from_views_bounded): the extents of both buffers are stated, a slot that reaches outside them is skipped and
        counted -> a third result, an int32 tensor of one element (read it after synchronize()); False: strsim_column_from_views."""
        import torch
        rows = views.numel() // 16
        off = torch.empty(rows + 1, dtype=torch.int32, device=views.device)
        val = torch.zeros(int(packed_bytes) + 64, dtype=torch.uint8, device=views.device)
        if not bounded:
            check(lib().strsim_column_from_views(self._h, views.data_ptr(), rows, long_values.data_ptr() if long_values is not None else None,
                                                 off.data_ptr(), val.data_ptr()))
            return off, val
        bad = torch.zeros(1, dtype=torch.int32, device=views.device)
        check(lib().strsim_column_from_views_bounded(self._h, views.data_ptr(), rows,
                                                     long_values.data_ptr() if long_values is not None else None,
                                                     long_values.numel() if long_values is not None else 0,
                                                     off.data_ptr(), val.data_ptr(), val.numel(), bad.data_ptr()))
        return off, val, bad

    def retire_oldest(self):
        """Retire the oldest pending call only (the caller knows by an event of its own that it has completed)."""
        check(lib().strsim_ctx_retire_oldest(self._h))

    def pairs_device_all(self, a_offsets, a_values, b_offsets, b_values, outs=None):
        """All five measures in one fused call -> list of five f64 tensors indexed like MEASURES."""
        import torch
        ra, rb = a_offsets.numel() - 1, b_offsets.numel() - 1
        n = rb if ra == 1 else ra
        if outs is None:
            outs = [torch.empty(n, dtype=torch.float64, device=a_offsets.device) for _ in range(5)]
        arr = (C.c_void_p * 5)(*[o.data_ptr() for o in outs])
        check(lib().strsim_pairs_device_all(self._h, a_offsets.data_ptr(), a_values.data_ptr(), ra,
                                            b_offsets.data_ptr(), b_values.data_ptr(), rb, arr, n))
        return outs

    # ---- host-resident (numpy) ---------------------------------------------------------------------
    def pairs_host(self, measure, a_offsets, a_values, b_offsets, b_values):
        """Synchronous: numpy uint32 offsets + uint8 values in, numpy f64 out."""
        ao = np.ascontiguousarray(a_offsets, dtype=np.uint32)
        bo = np.ascontiguousarray(b_offsets, dtype=np.uint32)
        av = np.ascontiguousarray(a_values, dtype=np.uint8)
        bv = np.ascontiguousarray(b_values, dtype=np.uint8)
        if av.size == 0:
            av = np.zeros(1, dtype=np.uint8)
        if bv.size == 0:
            bv = np.zeros(1, dtype=np.uint8)
        ra, rb = ao.size - 1, bo.size - 1
        n = rb if ra == 1 else ra
        if ra != rb and ra != 1 and rb != 1:
            n = 0
        out = np.empty(n, dtype=np.float64)
        check(lib().strsim_pairs_host(self._h, measure_id(measure), ao.ctypes.data, av.ctypes.data, ra,
                                      bo.ctypes.data, bv.ctypes.data, rb, out.ctypes.data, n))
        return out


class Codec:
    """Lossless 16-bit transport codec for one measure's result column (include/strsim_amd.h: strsim_codec_*)."""
    EXC_CAP = 1 << 20

    def __init__(self, ctx, measure, max_chars=32):
        import torch
        self.ctx = ctx
        self._h = C.c_void_p()
        check(lib().strsim_codec_create(ctx._h, measure_id(measure), int(max_chars), C.byref(self._h)))
        dev = torch.device("cuda", ctx.device)
        self.exc_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.exc_rows = torch.empty(self.EXC_CAP, dtype=torch.int32, device=dev)
        self.exc_vals = torch.empty(self.EXC_CAP, dtype=torch.float64, device=dev)

    @property
    def entries(self):
        return int(lib().strsim_codec_entries(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().strsim_codec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _exc(self, exc):
        """(count ptr, rows ptr, vals ptr, cap) of an exception block: this codec's own, or `exc` = (count, rows, vals)
        tensors of the caller (e.g. views into a buffer that is shipped together with the codes)."""
        if exc is None:
            return self.exc_count.data_ptr(), self.exc_rows.data_ptr(), self.exc_vals.data_ptr(), self.EXC_CAP
        cnt, rows, vals = exc
        return cnt.data_ptr(), rows.data_ptr(), vals.data_ptr(), min(rows.numel(), vals.numel())

    def encode(self, vals, codes=None, ctx=None, exc=None):
        """f64 tensor -> int16 tensor of codes (0xFFFF = exception, see exc_*); asynchronous on ctx's stream."""
        import torch
        ctx = ctx or self.ctx
        if codes is None:
            codes = torch.empty(vals.numel(), dtype=torch.int16, device=vals.device)
        pc, pr, pv, cap = self._exc(exc)
        check(lib().strsim_codec_encode(ctx._h, self._h, vals.data_ptr(), vals.numel(), codes.data_ptr(), pc, pr, pv, cap))
        return codes

    def decode(self, codes, out=None, ctx=None):
        import torch
        ctx = ctx or self.ctx
        if out is None:
            out = torch.empty(codes.numel(), dtype=torch.float64, device=codes.device)
        check(lib().strsim_codec_decode(ctx._h, self._h, codes.data_ptr(), codes.numel(), out.data_ptr()))
        return out

    @property
    def bits(self):
        """Bits per row of the packed transport (2^bits > entries; the all-ones code is the escape)."""
        return int(lib().strsim_codec_bits(self._h))

    def packed_words(self, n):
        return int(lib().strsim_codec_packed_words(self._h, int(n)))

    def encode_packed(self, vals, words=None, ctx=None, exc=None):
        """f64 tensor -> int64 tensor of packed_words(n) words, 64 // bits codes each; asynchronous on ctx's stream."""
        import torch
        ctx = ctx or self.ctx
        if words is None:
            words = torch.empty(self.packed_words(vals.numel()), dtype=torch.int64, device=vals.device)
        pc, pr, pv, cap = self._exc(exc)
        check(lib().strsim_codec_encode_packed(ctx._h, self._h, vals.data_ptr(), vals.numel(), words.data_ptr(), pc, pr, pv, cap))
        return words

    def decode_packed(self, words, n, out=None, ctx=None):
        import torch
        ctx = ctx or self.ctx
        if out is None:
            out = torch.empty(int(n), dtype=torch.float64, device=words.device)
        check(lib().strsim_codec_decode_packed(ctx._h, self._h, words.data_ptr(), int(n), out.data_ptr()))
        return out

    def patch_indirect(self, out, row_base, exc, overflow, ctx=None):
        """out[row_base + rows[i]] = vals[i] for i < min(count, cap), the count read on the device; `exc` = (count, rows, vals)
        tensors (an exception block as it arrived from another rank), `overflow`: int32 device tensor, incremented when
        count > cap."""
        ctx = ctx or self.ctx
        cnt, rows, vals = exc
        check(lib().strsim_codec_patch_indirect(ctx._h, out.data_ptr(), int(row_base), cnt.data_ptr(), rows.data_ptr(),
                                                vals.data_ptr(), min(rows.numel(), vals.numel()), overflow.data_ptr()))

    def decode_gathered(self, buf, seg_stride_bytes, nseg, chunk_rows, last_rows, packed, code_bytes, exc_cap, out, overflow, ctx=None,
                        first_seg=0):
        """The root's decode of every rank's segment of a gathered buffer (codes + exception block each) in ONE launch:
        strsim_codec_decode_gathered; `first_seg` = 1 leaves the root's own segment alone (strsim_codec_decode_gathered_from: the
        root copies its f64 shard in instead of coding and decoding it)."""
        ctx = ctx or self.ctx
        check(lib().strsim_codec_decode_gathered_from(ctx._h, self._h, buf.data_ptr(), int(seg_stride_bytes), int(first_seg), int(nseg),
                                                      int(chunk_rows), int(last_rows), 1 if packed else 0, int(code_bytes), int(exc_cap),
                                                      out.data_ptr(), overflow.data_ptr()))

    def patch(self, out, row_base, exc_rows, exc_vals, count, ctx=None):
        ctx = ctx or self.ctx
        check(lib().strsim_codec_patch(ctx._h, out.data_ptr(), int(row_base), exc_rows.data_ptr(), exc_vals.data_ptr(),
                                       int(count)))


def pack_strings(strings):
    """list[str|bytes] -> (uint32 offsets[n+1], uint8 values): the device column layout."""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
    offs = np.zeros(len(bs) + 1, dtype=np.uint32)
    if bs:
        offs[1:] = np.cumsum([len(x) for x in bs], dtype=np.uint64).astype(np.uint32)
    vals = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return offs, vals
