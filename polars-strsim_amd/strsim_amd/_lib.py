"""ctypes binding of libpolars_strsim_amd.so (C ABI: include/strsim_amd.h).

The library is built in-tree by polars-strsim_amd/Makefile into the drop-in package directory
polars_strsim/ (where Polars looks for the plugin).  There is no Python or CPU fallback: if the
library is missing this module raises.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# STRSIM_AMD_LIB: another build of the same library (A/B runs of compile-time variants on one box)
LIB_PATH = os.environ.get("STRSIM_AMD_LIB") or os.path.join(PKG_DIR, "polars_strsim", "libpolars_strsim_amd.so")

MEASURES = ("levenshtein", "jaro", "jaro_winkler", "jaccard", "sorensen_dice")  # strsim.rs:9-15
MEASURE_ID = {m: i for i, m in enumerate(MEASURES)}

STATUS = {0: "OK", 1: "ERR_SHAPE", 2: "ERR_ARG", 3: "ERR_NO_DEVICE", 4: "ERR_HIP", 5: "ERR_OOM", 6: "ERR_DTYPE",
          7: "ERR_INTERNAL", 8: "ERR_EARLIER_CALL"}


class StrsimError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{STATUS.get(code, code)}: {message}")
        self.code = code
        self.message = message


class ShapeMismatch(StrsimError, ValueError):
    """reference: PolarsError::ShapeMismatch, strsim.rs:48-52"""


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    torch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, needed by libtorch_hip.so as plain
    "libamdhip64.so").  Loaded first, this library binds /opt/rocm's copy; a later `import torch` then maps the bundled
    copy NEXT to it (its DT_NEEDED name matches neither the path nor the SONAME of the one already there), and the second
    runtime to initialise reports "No HIP GPUs are available".  Mapping torch's copy before ours makes the loader resolve
    our DT_NEEDED libamdhip64.so.7 by SONAME to it, and torch finds the same file already mapped.  A process without
    torch (the Polars plugin route) is not affected and runs on /opt/rocm's runtime.
    """
    import sys
    if "torch" in sys.modules or os.environ.get("STRSIM_KEEP_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {PKG_DIR}` (hipcc, gfx950). "
            "polars-strsim_amd has no Python/CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.strsim_abi_version.restype = C.c_uint32
    L.strsim_abi_version.argtypes = []
    L.strsim_last_error_message.restype = C.c_char_p
    L.strsim_last_error_message.argtypes = []
    L.strsim_device_count.restype = i32
    L.strsim_device_count.argtypes = []
    L.strsim_ctx_create.restype = i32
    L.strsim_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.strsim_ctx_destroy.restype = None
    L.strsim_ctx_destroy.argtypes = [vp]
    L.strsim_ctx_stream.restype = vp
    L.strsim_ctx_stream.argtypes = [vp]
    for name in ("strsim_pairs_device", "strsim_pairs_host"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, i32, vp, vp, u64, vp, vp, u64, vp, u64]
    L.strsim_pairs_device_all.restype = i32
    L.strsim_pairs_device_all.argtypes = [vp, vp, vp, u64, vp, vp, u64, C.POINTER(vp), u64]
    L.strsim_codec_create.restype = i32
    L.strsim_codec_create.argtypes = [vp, i32, C.c_uint32, C.POINTER(vp)]
    L.strsim_codec_destroy.restype = None
    L.strsim_codec_destroy.argtypes = [vp]
    L.strsim_codec_entries.restype = C.c_uint32
    L.strsim_codec_entries.argtypes = [vp]
    L.strsim_codec_encode.restype = i32
    L.strsim_codec_encode.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, C.c_uint32]
    L.strsim_codec_decode.restype = i32
    L.strsim_codec_decode.argtypes = [vp, vp, vp, u64, vp]
    L.strsim_codec_bits.restype = C.c_uint32
    L.strsim_codec_bits.argtypes = [vp]
    L.strsim_codec_packed_words.restype = u64
    L.strsim_codec_packed_words.argtypes = [vp, u64]
    L.strsim_codec_encode_packed.restype = i32
    L.strsim_codec_encode_packed.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, C.c_uint32]
    L.strsim_codec_decode_packed.restype = i32
    L.strsim_codec_decode_packed.argtypes = [vp, vp, vp, u64, vp]
    L.strsim_codec_patch.restype = i32
    L.strsim_codec_patch.argtypes = [vp, vp, u64, vp, vp, C.c_uint32]
    L.strsim_offsets_from_lengths.restype = i32
    L.strsim_offsets_from_lengths.argtypes = [vp, vp, u64, vp]
    L.strsim_ctx_retire_oldest.restype = i32
    L.strsim_ctx_retire_oldest.argtypes = [vp]
    L.strsim_codec_patch_indirect.restype = i32
    L.strsim_codec_patch_indirect.argtypes = [vp, vp, u64, vp, vp, vp, C.c_uint32, vp]
    L.strsim_ctx_synchronize.restype = i32
    L.strsim_ctx_synchronize.argtypes = [vp]
    L.strsim_split_offsets.restype = None
    L.strsim_split_offsets.argtypes = [u64, u64, vp]
    L.strsim_ctx_timing_enable.restype = i32
    L.strsim_ctx_timing_enable.argtypes = [vp, i32]
    L.strsim_ctx_timing_read.restype = i32
    L.strsim_ctx_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(C.c_double),
                                         C.POINTER(u64)]
    L.strsim_ctx_last_wave_rows.restype = u64
    L.strsim_ctx_last_wave_rows.argtypes = [vp]
    L.strsim_ctx_last_long_rows.restype = u64
    L.strsim_ctx_last_long_rows.argtypes = [vp]
    L.strsim_column_from_views.restype = i32
    L.strsim_column_from_views.argtypes = [vp, vp, u64, vp, vp, vp]
    L.strsim_column_from_views_bounded.restype = i32
    L.strsim_column_from_views_bounded.argtypes = [vp, vp, u64, vp, u64, vp, vp, u64, vp]
    L.strsim_ctx_last_late_rows.restype = u64
    L.strsim_ctx_last_late_rows.argtypes = [vp]
    L.strsim_codec_decode_gathered.restype = i32
    L.strsim_codec_decode_gathered.argtypes = [vp, vp, vp, u64, C.c_uint32, u64, u64, i32, u64, C.c_uint32, vp, vp]
    L.strsim_gather_unique_id.restype = i32
    L.strsim_gather_unique_id.argtypes = [vp]
    L.strsim_gather_create.restype = i32
    L.strsim_gather_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.strsim_gather_f64.restype = i32
    L.strsim_gather_f64.argtypes = [vp, vp, vp, u64, i32]
    L.strsim_ctx_get_stream_ordered.restype = i32
    L.strsim_ctx_get_stream_ordered.argtypes = [vp]
    L.strsim_gather_f64_ranges.restype = i32
    L.strsim_gather_f64_ranges.argtypes = [vp, vp, vp, vp, i32]
    L.strsim_gather_comm_count.restype = i32
    L.strsim_gather_comm_count.argtypes = [vp, C.POINTER(i32)]
    L.strsim_gather_destroy.restype = None
    L.strsim_gather_destroy.argtypes = [vp]
    L.strsim_codec_decode_gathered_from.restype = i32
    L.strsim_codec_decode_gathered_from.argtypes = [vp, vp, vp, u64, C.c_uint32, C.c_uint32, u64, u64, i32, u64, C.c_uint32, vp, vp]
    L.strsim_compact_segments.restype = i32
    L.strsim_compact_segments.argtypes = [vp, vp, vp, vp, vp, vp, i32]
    L.strsim_ctx_enqueued_ops.restype = u64
    L.strsim_ctx_enqueued_ops.argtypes = [vp]
    L.strsim_ctx_set_stream_ordered.restype = i32
    L.strsim_ctx_set_stream_ordered.argtypes = [vp, i32]
    _lib = L
    return L


def check(rc):
    if rc == 0:
        return
    msg = lib().strsim_last_error_message().decode("utf-8", "replace")
    if rc == 1:
        raise ShapeMismatch(rc, msg)
    raise StrsimError(rc, msg)


def measure_id(measure):
    if isinstance(measure, str):
        return MEASURE_ID[measure]
    return int(measure)
