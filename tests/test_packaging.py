"""the drop-in package builds the way a user installs it -- `pip wheel polars-strsim_amd/` runs the Makefile and
puts libpolars_strsim_amd.so INSIDE the polars_strsim package directory, where Polars scans for the plugin library
(reference polars_strsim/__init__.py:11-16 passes plugin_path = the package directory; reference pyproject.toml:1-34)."""
import os
import shutil
import subprocess
import sys
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wheel_carries_the_plugin_library(tmp_path):
    src = os.path.join(ROOT, "polars-strsim_amd")
    r = subprocess.run([sys.executable, "-m", "pip", "wheel", src, "--no-build-isolation", "--no-deps", "-w", str(tmp_path)],
                       capture_output=True, text=True, timeout=1200)
    for litter in ("build", "polars_strsim_amd.egg-info"):  # pip builds in the source tree: leave it as it was
        shutil.rmtree(os.path.join(src, litter), ignore_errors=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    wheels = [f for f in os.listdir(tmp_path) if f.endswith(".whl")]
    assert len(wheels) == 1 and wheels[0].startswith("polars_strsim_amd-0.2.3")
    names = zipfile.ZipFile(os.path.join(tmp_path, wheels[0])).namelist()
    for want in ("polars_strsim/__init__.py", "polars_strsim/utils.py", "polars_strsim/libpolars_strsim_amd.so",
                 "strsim_amd/__init__.py"):
        assert want in names, (want, names)
    # the wrappers keep the reference's names (polars_strsim/__init__.py:63-69) -- checked without importing polars
    text = open(os.path.join(src, "polars_strsim", "__init__.py")).read()
    for fn in ("levenshtein", "jaro", "jaro_winkler", "jaccard", "sorensen_dice"):
        assert f"def {fn}(expr: IntoExpr, other: IntoExpr) -> pl.Expr" in text
    assert "is_elementwise=True" in text
