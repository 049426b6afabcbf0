import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "polars-strsim_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _duplicate_test_names(path):
    """Top-level `def test_*` / `class Test*` names defined more than once in a file.  Python keeps the LAST definition, so the
    earlier test silently never runs and pytest cannot see it (tests/test_gpu_parity.py carried one for a round)."""
    import ast
    seen, dup = set(), []
    for node in ast.parse(open(path, encoding="utf-8").read(), filename=path).body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) and node.name.lower().startswith("test"):
            if node.name in seen:
                dup.append("%s:%d %s" % (os.path.basename(path), node.lineno, node.name))
            seen.add(node.name)
    return dup


def pytest_collectstart(collector):
    # (a collection ERROR, not a warning: the run fails until the shadowed test is given a name of its own)
    path = str(getattr(collector, "path", "") or "")
    if isinstance(collector, pytest.Module) and os.path.basename(path).startswith("test_"):
        dup = _duplicate_test_names(path)
        if dup:
            raise pytest.UsageError("duplicate top-level test names (the earlier definition is dead code): " + ", ".join(dup))


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# The driver runs `pytest -x`: the first failure hides everything collected behind it.  So the drop-in boundary runs first
# (C-ABI symbols, then the whole Polars-plugin ABI on the GPU, then the thin-ABI parity suite), the cheap suites next, and
# the minutes-long full-size / fuzz files last -- whatever the alphabet says.
_FILE_ORDER = ["test_abi_symbols.py", "test_plugin_abi_gpu.py", "test_plugin_configs_gpu.py", "test_hip_runtime_sharing.py",
               "test_gpu_parity.py", "test_gpu_multirank_smoke.py", "test_gather_abi.py", "test_packaging.py", "test_installed_wheel_gpu.py"]
_FILE_LAST = ["test_knobs_gpu.py", "test_gpu_hypothesis.py", "test_gpu_fuzz.py", "test_gpu_fullsize.py"]


def _file_rank(item):
    name = os.path.basename(str(item.fspath))
    if name in _FILE_ORDER:
        return _FILE_ORDER.index(name)
    if name in _FILE_LAST:
        return 1000 + _FILE_LAST.index(name)
    return 500


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)  # (stable: the order inside a file is kept)
    # `-m gpu` on a box without a GPU must fail loudly, not skip; plain runs skip gpu tests without a GPU.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _address_checks_of_the_lab_build(request):
    """With the bounds-checked lab build selected (STRSIM_AMD_LIB = a library made with EXTRA="-DSTRSIM_LAB -DSTRSIM_BOUNDS",
    bench_support/jobs/r5_bounds_fuzz.sh) every GPU test also asserts that no kernel formed an address outside its launch's extents.
    The product library has no such symbols: nothing happens."""
    yield
    if "gpu" not in request.keywords or not os.environ.get("STRSIM_AMD_LIB"):
        return
    import ctypes as C
    import strsim_amd as S
    L = S.lib()
    if not hasattr(L, "strsim_debug_bounds_kernels"):
        return
    import torch
    torch.cuda.synchronize()
    for unit in ("kernels", "codec"):
        f = getattr(L, "strsim_debug_bounds_" + unit)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        rec = (C.c_ulonglong * 6)()
        assert f(rec) == 0
        assert rec[0] == 0, "address out of bounds in the %s unit: hits %d, kernel %d site %d, row %d, value %#x not in [%#x, %#x]" % (
            unit, rec[0], rec[1] >> 32, rec[1] & 0xFFFFFFFF, rec[2], rec[3], rec[4], rec[5])
