"""A minimal stand-in for the Polars engine's side of the plugin ABI, built on pyarrow + ctypes.

It exports pyarrow string arrays (Utf8View "vu" like Polars, or "u"/"U") over the Arrow C Data Interface
as polars-ffi `SeriesExport`s, calls `_polars_plugin_<name>` in libpolars_strsim_amd.so exactly the way
polars-plan does (inputs owned by the callee, `return_value` pre-zeroed, error text fetched from
`_polars_plugin_get_last_error_message`), and imports the returned Float64 series.  Used by the tests and
by anything that wants the reference's operator surface without Polars installed.
"""
import ctypes as C

import pyarrow as pa

from ._lib import lib as _load


class ArrowSchema(C.Structure):
    pass


class ArrowArray(C.Structure):
    pass


ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                        ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))),
                        ("dictionary", C.POINTER(ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p)]
ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                       ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
                       ("children", C.POINTER(C.POINTER(ArrowArray))), ("dictionary", C.POINTER(ArrowArray)),
                       ("release", C.c_void_p), ("private_data", C.c_void_p)]


class SeriesExport(C.Structure):
    pass


RELEASE_SERIES = C.CFUNCTYPE(None, C.POINTER(SeriesExport))
SeriesExport._fields_ = [("field", C.POINTER(ArrowSchema)), ("arrays", C.POINTER(C.POINTER(ArrowArray))),
                         ("len", C.c_size_t), ("release", RELEASE_SERIES), ("private_data", C.c_void_p)]


class CallerContext(C.Structure):
    _fields_ = [("bitflags", C.c_uint64)]


RELEASE_SCHEMA = C.CFUNCTYPE(None, C.POINTER(ArrowSchema))
RELEASE_ARRAY = C.CFUNCTYPE(None, C.POINTER(ArrowArray))


class PluginError(RuntimeError):
    """What Polars raises as ComputeError("the plugin failed with message: ...")."""


class _Exported:
    """Keeps the C structs of one exported Series alive until the callee releases them."""

    def __init__(self, name, chunks, dtype):
        self.schema = ArrowSchema()
        pa.field(name, dtype)._export_to_c(C.addressof(self.schema))
        self.arrays = [ArrowArray() for _ in chunks]
        for st, ch in zip(self.arrays, chunks):
            ch._export_to_c(C.addressof(st))
        self.ptrs = (C.POINTER(ArrowArray) * max(len(chunks), 1))(*[C.pointer(a) for a in self.arrays])
        self.released = 0

        def _release(e):
            # polars-ffi c_release_series_export: drop the schema, free the boxes -- NOT the arrays
            self.released += 1
            if self.schema.release:
                RELEASE_SCHEMA(self.schema.release)(C.pointer(self.schema))
            e.contents.release = RELEASE_SERIES()
            e.contents.private_data = None

        self._cb = RELEASE_SERIES(_release)

    def fill(self, se):
        se.field = C.pointer(self.schema)
        se.arrays = C.cast(self.ptrs, C.POINTER(C.POINTER(ArrowArray)))
        se.len = len(self.arrays)
        se.release = self._cb
        se.private_data = 1

    def arrays_released(self):
        return all(not a.release for a in self.arrays)


def _chunks(x, layout):
    """-> (list of pyarrow arrays of the requested string layout, dtype)"""
    dtype = {"vu": pa.string_view(), "u": pa.string(), "U": pa.large_string()}[layout]
    if isinstance(x, pa.ChunkedArray):
        chunks = list(x.chunks)
    elif isinstance(x, pa.Array):
        chunks = [x]
    elif isinstance(x, (str, bytes)) or x is None:
        chunks = [pa.array([x], type=pa.string())]
    else:
        chunks = [pa.array(list(x), type=pa.string())]
    out = []
    for c in chunks:
        if pa.types.is_string(c.type) or pa.types.is_large_string(c.type) or pa.types.is_string_view(c.type):
            out.append(c if c.type == dtype else c.cast(dtype))
        else:
            return chunks, chunks[0].type  # non-string input: exported as is (exercises the dtype error)
    return out, dtype


def call_plugin(name, a, b, layout="vu", names=("a", "b"), parallel=False, _probe=None):
    """Evaluate plugin expression `name` over two columns (lists / pyarrow arrays / a single literal).
    Returns a pyarrow ChunkedArray of float64.  `layout` picks the Arrow string layout sent to the plugin."""
    L = _load()
    fn = getattr(L, "_polars_plugin_" + name)
    fn.restype = None
    fn.argtypes = [C.POINTER(SeriesExport), C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(SeriesExport),
                   C.POINTER(CallerContext)]
    L._polars_plugin_get_last_error_message.restype = C.c_char_p
    exported = []
    inputs = (SeriesExport * 2)()
    for i, x in enumerate((a, b)):
        chunks, dtype = _chunks(x, layout if isinstance(layout, str) else layout[i])
        ex = _Exported(names[i], chunks, dtype)
        ex.fill(inputs[i])
        exported.append(ex)
    ret = SeriesExport()  # SeriesExport::empty()
    ctx = CallerContext(1 if parallel else 0)
    fn(inputs, 2, None, 0, C.byref(ret), C.byref(ctx))
    if _probe is not None:
        _probe["series_released"] = [e.released for e in exported]
        _probe["arrays_released"] = [e.arrays_released() for e in exported]
    if not ret.private_data:  # SeriesExport::is_null()
        raise PluginError("the plugin failed with message: " + L._polars_plugin_get_last_error_message().decode())
    try:
        out = []
        fld = pa.Field._import_from_c(C.addressof(ret.field.contents)) if False else None
        for i in range(ret.len):
            out.append(pa.Array._import_from_c(C.addressof(ret.arrays[i].contents), pa.float64()))
        name_out = ret.field.contents.name.decode() if ret.field.contents.name else ""
    finally:
        ret.release(C.byref(ret))
    res = pa.chunked_array(out, type=pa.float64())
    if _probe is not None:
        _probe["name"] = name_out
    return res


def field_plugin(name, input_names=("a", "b")):
    """Planning-time output field of plugin expression `name` -> (field name, pyarrow type)."""
    L = _load()
    fn = getattr(L, "_polars_plugin_field_" + name)
    fn.restype = None
    fn.argtypes = [C.POINTER(ArrowSchema), C.c_size_t, C.POINTER(ArrowSchema)]
    fields = (ArrowSchema * len(input_names))()
    for i, n in enumerate(input_names):
        pa.field(n, pa.string_view())._export_to_c(C.addressof(fields[i]))
    out = ArrowSchema()
    fn(fields, len(input_names), C.byref(out))
    f = pa.Field._import_from_c(C.addressof(out))
    for i in range(len(input_names)):
        if fields[i].release:
            RELEASE_SCHEMA(fields[i].release)(C.pointer(fields[i]))
    return f.name, f.type


def plugin_version():
    L = _load()
    L._polars_plugin_get_version.restype = C.c_uint32
    v = L._polars_plugin_get_version()
    return v >> 16, v & 0xFFFF
