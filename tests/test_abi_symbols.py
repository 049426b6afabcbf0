"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the headers
declare, and its no-compute paths (version, field resolution, shape/dtype errors, loud failure without a
GPU) behave like the reference's (src/expressions/mod.rs:8-31, strsim.rs:46-52)."""
import ctypes as C
import os
import re

import pyarrow as pa
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    h1 = open(os.path.join(ROOT, "include", "strsim_amd.h")).read()
    names |= set(re.findall(r"STRSIM_API[^;(]*?\b(strsim_\w+)\s*\(", h1))
    h2 = open(os.path.join(ROOT, "include", "polars_plugin_abi.h")).read()
    names |= set(re.findall(r"POLARS_PLUGIN_API[^;(]*?\b(_polars_plugin_\w+)\s*\(", h2))
    for m in re.findall(r"^POLARS_PLUGIN_DECLARE\((\w+)\)", h2, flags=re.M):
        names.add("_polars_plugin_" + m)
        names.add("_polars_plugin_field_" + m)
    names.discard("_polars_plugin_")
    return names


def test_library_exports_every_declared_symbol():
    import strsim_amd
    L = strsim_amd.lib()  # not a bare CDLL: the binding keeps the process on one HIP runtime (strsim_amd/_lib.py)
    syms = declared_symbols()
    assert len(syms) >= 13 + 12
    for s in sorted(syms):
        assert hasattr(L, s), f"{s} declared in include/ but not exported"


def test_versions():
    import strsim_amd
    from strsim_amd import arrow_host as H
    v = strsim_amd.lib().strsim_abi_version()
    assert v >> 16 == 1 and (v & 0xFFFF) >= 4  # 1.4: one-launch calls are opt-in (strsim_ctx_set_stream_ordered(ctx, 0))
    assert H.plugin_version() == (0, 1)


@pytest.mark.parametrize("name", ["levenshtein", "jaro", "jaro_winkler", "jaccard", "sorensen_dice"])
def test_output_field_is_float64_named_after_first_input(name):
    from strsim_amd import arrow_host as H
    assert H.field_plugin(name, ("name_a", "name_b")) == ("name_a", pa.float64())  # output_type=Float64, mod.rs:8


def test_shape_mismatch_message_and_input_ownership():
    from strsim_amd import arrow_host as H
    probe = {}
    with pytest.raises(H.PluginError, match="Inputs must have the same length, or one of them must be a Utf8 literal."):
        H.call_plugin("jaro", ["a", "b"], ["a", "b", "c"], _probe=probe)  # strsim.rs:48-52
    assert probe["series_released"] == [1, 1] and probe["arrays_released"] == [True, True]


def test_dtype_error():
    from strsim_amd import arrow_host as H
    probe = {}
    with pytest.raises(H.PluginError, match="invalid series dtype: expected `String`"):
        H.call_plugin("levenshtein", pa.array([1, 2]), pa.array(["a", "b"]), _probe=probe)  # `.str()?`, strsim.rs:46-47
    assert probe["series_released"] == [1, 1] and probe["arrays_released"] == [True, True]


def test_split_offsets_matches_reference_rule():
    import strsim_amd as S
    assert S.split_offsets(10, 1) == [(0, 10)]
    assert S.split_offsets(10, 3) == [(0, 3), (3, 3), (6, 4)]  # strsim.rs:25-35
    assert S.split_offsets(2, 4) == [(0, 0), (0, 0), (0, 0), (0, 2)]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import strsim_amd as S
    from strsim_amd import arrow_host as H
    assert S.device_count() == 0
    with pytest.raises(S.StrsimError, match="no HIP device"):
        S.Context(0)
    with pytest.raises(H.PluginError, match="no HIP device"):
        H.call_plugin("levenshtein", ["a"], ["b"])
