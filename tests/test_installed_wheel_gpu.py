"""SURVEY 8 f4, the installed form: the wheel is built, pip-installed into an empty prefix, and a fresh interpreter that sees
ONLY that prefix (not the source tree) drives the `.so` it finds inside the installed `polars_strsim/` directory -- the
directory the reference passes as `plugin_path` (reference polars_strsim/__init__.py:11-16, pyproject.toml:1-34) -- through
the plugin ABI on the GPU.  The child prints its results; the parent compares them with the oracle."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import gen
import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import strsim_amd
from strsim_amd import arrow_host as H, _lib
prefix = sys.argv[1]
assert os.path.realpath(strsim_amd.__file__).startswith(os.path.realpath(prefix)), strsim_amd.__file__
so = os.path.realpath(_lib.LIB_PATH)
assert so == os.path.join(os.path.realpath(prefix), "polars_strsim", "libpolars_strsim_amd.so"), so
job = json.load(open(sys.argv[2]))
out = {"so": so, "version": list(H.plugin_version()), "field": [H.field_plugin("jaro", ("x", "y"))[0]]}
for m in job["measures"]:
    got = H.call_plugin(m, job["a"], job["b"])
    out[m] = [None if v is None else float.hex(v) for v in got.to_pylist()]
    out[m + "_lit"] = [None if v is None else float.hex(v) for v in H.call_plugin(m, job["a"], "phillips").to_pylist()]
json.dump(out, open(sys.argv[3], "w"))
"""


def test_installed_wheel_runs_the_plugin_from_site_packages(tmp_path):
    src = os.path.join(ROOT, "polars-strsim_amd")
    wh, prefix = tmp_path / "wheel", tmp_path / "prefix"
    r = subprocess.run([sys.executable, "-m", "pip", "wheel", src, "--no-build-isolation", "--no-deps", "-w", str(wh)],
                       capture_output=True, text=True, timeout=1800)
    for litter in ("build", "polars_strsim_amd.egg-info"):  # pip builds in the source tree: leave it as it was
        shutil.rmtree(os.path.join(src, litter), ignore_errors=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    wheel = [os.path.join(wh, f) for f in os.listdir(wh) if f.endswith(".whl")]
    assert len(wheel) == 1
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-deps", "--no-index", "--target", str(prefix), wheel[0]],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(prefix / "polars_strsim" / "libpolars_strsim_amd.so")

    A, B = gen.pairs(301, 5000, gen.ASCII_LOWER, 0, 40)
    A2, B2 = gen.pairs(302, 200, gen.MIXED, 0, 30)
    A, B = A + A2 + [None, "x"], B + B2 + ["y", None]
    job = tmp_path / "job.json"
    res = tmp_path / "res.json"
    json.dump({"measures": list(O.MEASURES), "a": A, "b": B}, open(job, "w"))
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "STRSIM_AMD_LIB")}
    env["PYTHONPATH"] = str(prefix)  # the installed copy only: the child never sees the source tree
    r = subprocess.run([sys.executable, "-c", CHILD, str(prefix), str(job), str(res)], capture_output=True, text=True,
                       timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.load(open(res))
    assert out["version"] == [0, 1] and out["field"] == ["x"]
    for m in O.MEASURES:
        for key, rhs in ((m, B), (m + "_lit", ["phillips"] * len(A))):
            got = out[key]
            assert len(got) == len(A)
            for i, (x, y, g) in enumerate(zip(A, rhs, got)):
                if x is None or y is None:
                    assert g is None, (key, i)
                else:
                    e = O.pair(m, x, y)
                    assert g is not None and np.float64(float.fromhex(g)).view(np.uint64) == np.float64(e).view(np.uint64), (key, i, x, y)
