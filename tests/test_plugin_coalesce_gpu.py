"""Small calls of concurrent engine threads can share a launch (opt-in; csrc/plugin_pipeline.h: Combiner; SURVEY 8 f3 "batching of concurrent
small calls"; reference: `is_elementwise=True`, polars_strsim/__init__.py:15 -- the engine calls per morsel / per group from its own
threads).  Every call's result must be what it would have been alone: bit for bit the oracle's, nulls in place."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "helpers", "coalesce_child.py")


def _run(env_extra, threads=12, calls=60):
    env = {k: v for k, v in os.environ.items() if not k.startswith("POLARS_STRSIM_")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, CHILD, str(threads), str(calls)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_concurrent_small_calls_are_combined_and_stay_bit_exact():
    """Twelve threads x sixty small calls of mixed sizes and measures, frames with slow rows and nulls among them; the threshold is
    lowered to two calls in flight so that Python's threads reach it: launches are shared, nothing differs from the oracle."""
    d = _run({"POLARS_STRSIM_COALESCE": "1", "POLARS_STRSIM_COALESCE_MIN_INFLIGHT": "2"})
    assert d["bad"] == []
    assert d["combined_launches"] > 0 and d["calls_combined"] > d["combined_launches"] and d["most_calls_in_one_launch"] >= 2


def test_every_small_call_combined_even_alone():
    """Threshold one: every eligible call goes through the combiner, also when it is the only member of its batch."""
    d = _run({"POLARS_STRSIM_COALESCE": "1", "POLARS_STRSIM_COALESCE_MIN_INFLIGHT": "1"}, threads=3, calls=40)
    assert d["bad"] == [] and d["calls_combined"] >= 100


def test_off_by_default():
    """Opt-in (POLARS_STRSIM_COALESCE=1): by its own measurement the combiner does not pay on a 16-CPU box (profiles/r6_small_calls.txt)."""
    d = _run({"POLARS_STRSIM_COALESCE_MIN_INFLIGHT": "1"}, threads=4, calls=30)
    assert d["bad"] == [] and d["combined_launches"] == 0 and d["calls_on_the_ordinary_path"] >= 120
