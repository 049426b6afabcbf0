"""Loader for the plugin layer's host-side test hooks (tests/cpu_harness/libplugin_testhooks.so, `make -C polars-strsim_amd
testhooks`): the packer and the validity builder of csrc/polars_plugin.cpp, without a GPU.  Test infrastructure only -- the
product library does not export these entry points."""
import ctypes as C
import os
import subprocess

import numpy as np

from strsim_amd import arrow_host as H
from strsim_amd.arrow_host import PluginError  # (what the hooks raise)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "cpu_harness", "libplugin_testhooks.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("STRSIM_TESTHOOKS_LIB")  # (tests/run_sanitizers.sh: a sanitizer build of the hooks)
        if not so:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "polars-strsim_amd"), "testhooks"])
            so = SO
        H._load()  # (maps the HIP runtime the way the product does, then the product library the hooks link against)
        L = C.CDLL(so)
        L._strsim_test_pack_series.restype = C.c_int
        L._strsim_test_pack_series.argtypes = [C.POINTER(H.SeriesExport), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.c_uint]
        L._strsim_test_validity.restype = C.c_int
        L._strsim_test_validity.argtypes = [C.POINTER(H.SeriesExport), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                                            C.POINTER(C.c_uint64), C.c_uint]
        L._strsim_test_pack_onepass.restype = C.c_int
        L._strsim_test_pack_onepass.argtypes = [C.POINTER(H.SeriesExport), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_void_p,
                                                C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L._strsim_test_pack_views.restype = C.c_int
        L._strsim_test_pack_views.argtypes = [C.POINTER(H.SeriesExport), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_void_p,
                                              C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L._strsim_test_pack_grants.restype = C.c_int
        L._strsim_test_pack_grants.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(C.c_int)]
        L._polars_plugin_get_last_error_message.restype = C.c_char_p
        _lib = L
    return _lib


def pack_series(x, layout="vu", r0=0, r1=None, threads=1):
    """Run the plugin's host-side packer on one column: -> (offsets u32, values u8, validity bool, rows of the series)."""
    L = lib()
    chunks, dtype = H._chunks(x, layout)
    n = sum(len(c) for c in chunks)
    if r1 is None:
        r1 = n
    nbytes = sum(c.nbytes for c in chunks) + 64 * (n + 1)
    ex = H._Exported("col", chunks, dtype)
    se = H.SeriesExport()
    ex.fill(se)
    off = np.zeros(max(r1 - r0, 0) + 1, dtype=np.uint32)
    val = np.zeros(nbytes + 64, dtype=np.uint8)
    valid = np.zeros(max(r1 - r0, 0) + 1, dtype=np.uint8)
    rows, used = C.c_uint64(), C.c_uint64()
    rc = L._strsim_test_pack_series(C.byref(se), r0, r1, off.ctypes.data, val.ctypes.data, val.size, C.byref(rows), C.byref(used),
                                    valid.ctypes.data, threads)
    if rc != 0:
        raise H.PluginError(L._polars_plugin_get_last_error_message().decode())
    assert ex.released == 1 and ex.arrays_released()
    return off, val[: used.value], valid[: max(r1 - r0, 0)].astype(bool), rows.value


def pack_onepass(x, r0, r1, bytes_per_row, threads=1):
    """The one-pass packer on a view column: -> None when it gives up (overflow / a string beyond 255 bytes), else
    (lengths u8[r1 - r0], values u8 closed up, segments)."""
    L = lib()
    chunks, dtype = H._chunks(x, "vu")
    ex = H._Exported("col", chunks, dtype)
    se = H.SeriesExport()
    ex.fill(se)
    n = r1 - r0
    val = np.zeros(n * 256 + 4096, dtype=np.uint8)
    lens = np.zeros(n + 64, dtype=np.uint8)
    used, nseg = C.c_uint64(), C.c_int()
    rc = L._strsim_test_pack_onepass(C.byref(se), r0, r1, int(bytes_per_row * 256), threads, val.ctypes.data, val.size,
                                     lens.ctypes.data, C.byref(used), C.byref(nseg))
    assert ex.released == 1 and ex.arrays_released()
    if rc < 0:
        raise H.PluginError(L._polars_plugin_get_last_error_message().decode())
    if rc == 0:
        return None
    return lens[:n], val[: used.value], nseg.value


def pack_views(x, r0, r1, threads=1, long_bytes_per_row=None):
    """The view-native packer on a view column: -> (list of the rows' strings as bytes, rebuilt from the shipped views and the
    long-string area the way the device does, packed size).  long_bytes_per_row None: the segments are sized exactly."""
    L = lib()
    chunks, dtype = H._chunks(x, "vu")
    ex = H._Exported("col", chunks, dtype)
    se = H.SeriesExport()
    ex.fill(se)
    n = r1 - r0
    views = np.zeros((n + 8) * 16, dtype=np.uint8)
    lng = np.zeros(n * 320 + (1 << 20), dtype=np.uint8)
    span, total = C.c_uint64(), C.c_uint64()
    est = 0xFFFFFFFFFFFFFFFF if long_bytes_per_row is None else int(long_bytes_per_row * 256)
    rc = L._strsim_test_pack_views(C.byref(se), r0, r1, est, threads, views.ctypes.data, lng.ctypes.data, lng.size, C.byref(span),
                                   C.byref(total))
    assert ex.released == 1 and ex.arrays_released()
    if rc != 0:
        raise H.PluginError(L._polars_plugin_get_last_error_message().decode())
    v = views[: n * 16].reshape(n, 16)
    lens = v[:, :4].copy().view(np.uint32)[:, 0]
    offs = v[:, 12:16].copy().view(np.uint32)[:, 0]
    out = []
    for i in range(n):
        ln = int(lens[i])
        out.append(bytes(v[i, 4:4 + ln]) if ln <= 12 else bytes(lng[int(offs[i]):int(offs[i]) + ln]))
        if ln > 12:
            assert int(offs[i]) + ln <= span.value
    return out, total.value


def validity(a, b, layouts=("vu", "vu"), threads=1, vals=None):
    """The output validity the plugin builds for a call over columns a, b: -> (bool array of n rows, null count)."""
    L = lib()
    inputs = (H.SeriesExport * 2)()
    keep = []
    for i, x in enumerate((a, b)):
        chunks, dtype = H._chunks(x, layouts[i])
        ex = H._Exported("c%d" % i, chunks, dtype)
        ex.fill(inputs[i])
        keep.append(ex)
    n = max(sum(len(c) for c in H._chunks(a, layouts[0])[0]), sum(len(c) for c in H._chunks(b, layouts[1])[0]))
    words = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
    nulls, rows = C.c_int64(), C.c_uint64()
    rc = L._strsim_test_validity(inputs, words.ctypes.data, C.byref(nulls), vals.ctypes.data if vals is not None else None,
                                 C.byref(rows), threads)
    if rc != 0:
        raise H.PluginError(L._polars_plugin_get_last_error_message().decode())
    assert all(e.released == 1 and e.arrays_released() for e in keep)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[: rows.value].astype(bool)
    tail = np.unpackbits(words.view(np.uint8), bitorder="little")[rows.value: ((rows.value + 63) // 64) * 64]
    assert not tail.any()  # bits past the column are zero
    return bits, nulls.value


def pack_grants(engine_parallel, n_calls, rows):
    """Packing threads of n_calls calls in flight together (entered one after the other) -> (threads per call, helper threads
    lent out while they all run, helper threads lent out after they have all returned)."""
    t = np.zeros(n_calls, dtype=np.uint32)
    out = C.c_int()
    after = lib()._strsim_test_pack_grants(1 if engine_parallel else 0, n_calls, rows, t.ctypes.data, C.byref(out))
    return [int(x) for x in t], out.value, after
