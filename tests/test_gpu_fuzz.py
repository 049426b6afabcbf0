"""A short run of the differential fuzzer (tests/fuzz_gpu.py: random scripts x length classes x measures x literal
sides against the oracle, bit for bit).  Longer runs: `python tests/fuzz_gpu.py <seconds> <seed>`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 11])
def test_fuzz_short(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "12", str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "fuzz ok" in r.stdout
