"""bench.py's launcher logic, as far as it runs without a GPU: `--gpus N` with no WORLD_SIZE in the environment starts the ranks
itself (a child torch.distributed.run, never an exec), refuses N ranks on fewer than N GPUs before it starts anything, and
leaves with the child's exit code.  The run that succeeds (two gloo ranks on one GPU) is tests/test_gpu_multirank_smoke.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ENV = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def _bench(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=_ENV, capture_output=True, text=True, timeout=600)


def test_more_ranks_than_gpus_is_refused_before_any_rank_starts():
    import torch
    have = torch.cuda.device_count()
    r = _bench("--gpus", str(have + 2), "--rows", "1000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert f"but this node has {have} GPU(s)" in r.stderr and "starting" not in r.stderr


def test_ranks_are_started_as_children_and_their_exit_code_is_relayed():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("on a GPU box the run succeeds: tests/test_gpu_multirank_smoke.py")
    r = _bench("--gpus", "2", "--same-device", "--backend", "gloo", "--rows", "1000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert r.returncode != 0 and "bench.py needs a GPU" in r.stderr  # what each rank said; the launcher's code came back
