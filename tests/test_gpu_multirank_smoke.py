"""Two ranks on ONE GPU (gloo, host-staged collectives) exercise bench.py's N > 1 control flow end to end: shard
generation, kernels, side-stream shipping of every step's result shard (raw f64, 16-bit and bit-packed codec transport, single
measure and the fused five-measure pass) and the root-side decode -- rank 0 verifies what it gathered against every
rank's checksum.  RCCL itself needs one GPU per rank and is only exercised by the driver's multi-GPU run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra, rows="1000000"):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rows", rows, "--backend", "gloo", "--same-device", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("extra,transport", [([], "9-bit codes, packed"), (["--no-codec"], "f64"), (["--config", "cfg4"], "u16 codes"),
                                             (["--measure", "jaccard"], "10-bit codes, packed")])
def test_two_ranks_ship_and_verify(extra, transport):
    """Strong scaling (the default, = BASELINE's metric): the frame's rows are cut by split_offsets(rows, 2)."""
    d = _run(extra)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["rows_total"] == 1000000 and d["config"]["rows_rank0"] == 500000
    assert d["config"]["gather_transport"] == transport
    assert d["config"]["gather_verified"] is True
    assert d["config"]["codec_exceptions"] == 0
    # the line audits itself: the world torch.distributed formed, its backend, every rank's device
    dd = d["config"]["distributed"]
    assert dd["world_size"] == 2 and dd["backend"] == "gloo" and dd["same_device"] is True
    assert [x["rank"] for x in dd["devices"]] == [0, 1] and all(x["cuda_device"] == 0 for x in dd["devices"])


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment: bench.py starts the two ranks itself (a child
    torch.distributed.run), relays rank 0's JSON line and leaves with the child's exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--rows", "1000000",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "starting 2 ranks" in r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["distributed"]["world_size"] == 2 and d["config"]["gather_verified"] is True
    assert d["config"]["call_mode"].startswith("stream_ordered")


def test_bench_refuses_more_ranks_than_gpus_before_it_starts_any():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--rows", "1000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "but this node has" in (r.stderr + r.stdout)


def test_same_device_is_refused_with_rccl():
    """Two ranks on one GPU are a smoke test of the control flow, not a measurement: only with the gloo backend."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--same-device", "--backend", "nccl", "--rows", "1000",
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--same-device needs --backend gloo" in (r.stderr + r.stdout)


@pytest.mark.parametrize("extra", [[], ["--no-codec"]])
def test_ragged_shards(extra):
    """1 000 001 rows over 2 ranks: the last rank's shard is one row longer (strsim.rs:25-35)."""
    d = _run(extra, rows="1000001")
    assert d["config"]["rows_rank0"] == 500000 and d["config"]["gather_verified"] is True


def test_root_share_is_a_named_deviation():
    """--root-share 0.5: rank 0 (which also decodes the gathered column) holds half an equal share; the line says so, and what rank 0
    gathered still equals every rank's shard (an uneven partition: the segments are decoded one by one)."""
    d = _run(["--root-share", "0.5"], rows="1000000")
    assert d["config"]["rows_rank0"] == 249984 and d["config"]["partition"].startswith("DEVIATION")  # (whole 64-row chunks)
    assert d["config"]["gather_verified"] is True
    d = _run([], rows="1000000")
    assert d["config"]["partition"].startswith("split_offsets")


def test_weak_scaling_option():
    d = _run(["--scaling", "weak"], rows="300000")
    assert d["scaling"] == "weak" and d["config"]["rows_total"] == 600000 and d["config"]["gather_verified"] is True


def test_codec_exceptions_travel_and_are_patched():
    """A frame with strings of up to 128 bytes coded with the 32-character tables: every value outside the table (rows with
    a longer string) has to reach rank 0 through the exception block -- the gathered column must still equal the shards."""
    d = _run(["--config", "cfg3", "--force-codec"], rows="100000")
    assert d["config"]["gather_transport"].endswith("codes") or "codes" in d["config"]["gather_transport"]
    assert d["config"]["codec_exceptions"] > 1000          # the long rows really were exceptions ...
    assert d["config"]["gather_verified"] is True           # ... and arrived
