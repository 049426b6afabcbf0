"""The ONE JSON line of bench.py (the driver's contract): every required key, the two extra objects (roofline, cpu_baseline), the fields
added in round 6 and the arithmetic that ties them together -- on a small frame, so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ENV = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=_ENV, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_the_line_carries_the_contract_and_adds_up():
    d = _bench("--rows", "4000000", "--steps", "6", "--warmup", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = rows x steps / time
    assert abs(d["value"] - 4_000_000 / d["ms_per_step"] / 1e3) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_read_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02  # the kernel fits inside the step
    # the counters are never taken inside a bench run: a replayed figure names its source, a missing one says so
    assert isinstance(r["traffic_source"], str) and (r["traffic"] is None) == r["traffic_source"].startswith("none")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "rows" in c["sample"]
    assert d["parity_vs_oracle_on_sample"]["bit_mismatches"] == 0
    # round 6: the default call mode beside the headline's, one cold call, one call on an idle GPU, the mode read back from the context
    assert d["config"]["call_mode"].startswith("one_launch") and d["config"]["enqueued_kernels_and_copies_per_step"] == 1.0
    assert d["value_default_mode"] > 0 and d["default_mode"]["enqueued_kernels_and_copies_per_step"] == 5.0
    assert d["default_mode"]["call_mode"].startswith("stream_ordered")
    assert d["cold_first_call_ms"] > d["idle_gpu_call_ms"] > 0


def test_extra_modes_can_be_switched_off():
    d = _bench("--rows", "1000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-extra-modes")
    assert d["value_default_mode"] is None and d["cold_first_call_ms"] is None and d["idle_gpu_call_ms"] is None and "cpu_baseline" not in d
