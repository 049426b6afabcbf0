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


@pytest.mark.parametrize("extra,transport", [([], "9-bit codes, packed"), (["--no-codec"], "f64"), (["--config", "cfg4"], "u16 codes"), (["--measure", "jaccard"], "10-bit codes, packed")])
def test_two_ranks_ship_and_verify(extra, transport):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rows", "1000000", "--backend", "gloo", "--same-device", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["gather_transport"] == transport
    assert d["config"]["gather_verified"] is True
    assert d["config"]["codec_exceptions"] == 0
