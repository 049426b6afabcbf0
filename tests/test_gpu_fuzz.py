"""A short run of the differential fuzzer (tests/fuzz_gpu.py: random scripts x length classes x measures x literal
sides against the oracle, bit for bit).  Longer runs: `python tests/fuzz_gpu.py <seconds> <seed>`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz(seconds, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), str(seconds), str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "seed %d: %s" % (seed, (r.stdout + r.stderr)[-2000:])
    assert "fuzz ok" in r.stdout


@pytest.mark.parametrize("seed", [3, 11])
def test_fuzz_short(seed):
    _fuzz(12, seed)


def test_fuzz_fresh_seed():
    """One seed nobody has run before, from the wall clock (the fault that shipped inside round 4 was found by fresh
    seeds, not by the suite's fixed ones); a failure names the seed, `python tests/fuzz_gpu.py 60 <seed>` replays it."""
    import time
    seed = int(time.time()) % 1_000_000_007
    print("fresh fuzz seed:", seed)
    _fuzz(20, seed)
