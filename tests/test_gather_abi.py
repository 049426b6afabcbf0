"""The C ABI's gather of the result shards (include/strsim_amd.h: strsim_gather_*, csrc/strsim_gather.cpp) with MORE THAN ONE RANK
-- VERDICT r5 / ADVICE r5: until round 6 it had only ever run with world = 1, where no send or receive is posted.

  * CPU: the hand-written declarations of RCCL's ABI (csrc/strsim_rccl_abi.h) against the real <rccl/rccl.h>; the argument
    plumbing of strsim_amd.distributed.AbiGather with a stubbed library; gather_column with an explicit partition on two gloo ranks
    (--root-share with the f64 transport used to raise).
  * GPU, one device: 2 and 3 rank processes sharing the GPU over the tests' stand-in transport (tests/cpu_harness/fake_rccl.cpp,
    selected with STRSIM_RCCL_LIB -- real RCCL refuses two ranks on one device): ragged shards, root != 0, fewer rows than ranks,
    the explicit-ranges form; and bench.py --gather abi end to end the same way.
  * GPU, two devices (skipped on a one-GPU box): the same child processes over REAL RCCL.
"""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, "cpu_harness", "libfake_rccl.so")
CHILD = os.path.join(HERE, "helpers", "gather_child.py")


def _fake_rccl():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpu_harness")])
    assert os.path.exists(FAKE)
    return FAKE


def test_rccl_abi_declarations_match_the_header():
    """csrc/strsim_rccl_abi.h declares nine RCCL entry points by hand (the library neither includes rccl.h nor links librccl): sizes,
    alignment, arity and pointer-ness of every parameter, the by-value unique id and the two constants against the real header."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        "-I" + os.path.join(ROOT, "polars-strsim_amd", "csrc"), os.path.join(HERE, "cpu_harness", "rccl_abi_check.cpp")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_the_stand_in_transport_builds_against_the_real_header():
    lib = C.CDLL(_fake_rccl())
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd",
                 "ncclGetErrorString", "ncclCommCount"):
        assert hasattr(lib, name), name


class _StubLib:
    """Records what AbiGather hands to the C ABI (no GPU, no RCCL)."""
    def __init__(self):
        self.calls = []

    def strsim_gather_create(self, ctx, uid, world, rank, out):
        self.calls.append(("create", bytes(uid), world, rank))
        out._obj.value = 0xABC0
        return 0

    def strsim_gather_f64(self, h, shard, column, total, root):
        self.calls.append(("f64", shard, column, total, root))
        return 0

    def strsim_gather_f64_ranges(self, h, shard, column, flat, root):
        self.calls.append(("ranges", shard, column, list(flat), root))
        return 0

    def strsim_gather_comm_count(self, h, out):
        out._obj.value = 7
        return 0

    def strsim_gather_destroy(self, h):
        self.calls.append(("destroy",))


class _T:  # a tensor as far as AbiGather looks at one
    def __init__(self, ptr, n):
        self._p, self._n = ptr, n

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n


def test_abi_gather_argument_plumbing_with_a_stubbed_library(monkeypatch):
    import strsim_amd._lib as L
    from strsim_amd.distributed import AbiGather, shard_ranges
    stub = _StubLib()
    monkeypatch.setattr(L, "lib", lambda: stub)

    class Ctx:
        _h = C.c_void_p(1)
    g = AbiGather(Ctx(), bytes(range(128)), 4, 2)
    assert stub.calls[-1] == ("create", bytes(range(128)), 4, 2)
    g.gather(_T(0x1000, 25), None, total_rows=103, root=1)
    assert stub.calls[-1] == ("f64", 0x1000, None, 103, 1)
    g.gather(_T(0x1000, 0), _T(0x2000, 103), total_rows=103, root=2)  # an empty shard is a NULL pointer
    assert stub.calls[-1] == ("f64", None, 0x2000, 103, 2)
    parts = shard_ranges(1_000_000, 4, 0.5)
    g.gather(_T(0x1000, parts[2][1]), None, root=0, parts=parts)
    kind, shard, column, flat, root = stub.calls[-1]
    assert (kind, shard, column, root) == ("ranges", 0x1000, None, 0)
    assert flat == [v for p in parts for v in p] and flat[1] == 124992 and sum(flat[1::2]) == 1_000_000
    with pytest.raises(ValueError):
        g.gather(_T(0x1000, 1), None, root=0, parts=parts[:3])
    assert g.comm_count() == 7
    g.close()
    assert stub.calls[-1] == ("destroy",)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _parts_worker(rank, world, port, n_rows, share, q):
    for p in (os.path.join(ROOT, "polars-strsim_amd"), HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from strsim_amd.distributed import gather_column, shard_ranges
        parts = shard_ranges(n_rows, world, share)
        off, ln = parts[rank]
        local = torch.arange(off, off + ln, dtype=torch.float64) * 0.5
        ok = True
        try:  # without `parts` the shard of a deviating partition is refused (what ShardGatherer's f64 branch ran into)
            gather_column(local, n_rows, dst=0)
            ok = parts == shard_ranges(n_rows, world)
        except ValueError:
            pass
        # (every rank raised or none did: the partition is the same everywhere)
        full = gather_column(local, n_rows, dst=0, parts=parts)
        if rank == 0:
            q.put(ok and full.numel() == n_rows and bool((full == torch.arange(n_rows, dtype=torch.float64) * 0.5).all()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,share", [(100_003, 0.5), (100_003, 0.0), (64_000, 1.0)])
def test_gather_column_takes_an_explicit_partition(n_rows, share):
    """shard_ranges(..., root_share) + the f64 transport -- gather_column recomputed split_offsets and raised."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_parts_worker, args=(r, 2, port, n_rows, share, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


# ---- GPU -----------------------------------------------------------------------------------------------------------------------

def _ranks(world, rows, root, mode, tmp_path, devices, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    idf = str(tmp_path / "unique_id.bin")
    procs = [subprocess.Popen([sys.executable, CHILD, str(world), str(r), str(devices[r]), idf, str(rows), str(root), mode], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "GATHER-OK %d" % world in so, (r, so[-500:], se[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("world,rows,root,mode", [(2, 100_001, 0, "split"), (3, 70_001, 2, "split"), (3, 2, 0, "split"), (2, 60_000, 1, "split"),
                                                  (2, 100_000, 0, "ranges:0.5"), (3, 90_001, 1, "ranges:0.25")])
def test_c_abi_gather_with_several_ranks_sharing_one_gpu(world, rows, root, mode, tmp_path):
    """strsim_gather_f64 / strsim_gather_f64_ranges post the right sends and receives for N > 1: peers, counts, `column + offset`
    addresses, the ragged last shard, a root that is not rank 0, fewer rows than ranks, an explicit partition -- every rank computes its
    shard with the product kernels and the root holds the oracle's column bit for bit.  Transport: the tests' stand-in (see above)."""
    _ranks(world, rows, root, mode, tmp_path, [0] * world, {"STRSIM_RCCL_LIB": _fake_rccl()})


@pytest.mark.gpu
@pytest.mark.parametrize("world,rows,root,mode", [(2, 1_000_001, 0, "split"), (2, 1_000_001, 1, "ranges:0.5")])
def test_c_abi_gather_over_real_rccl_with_one_gpu_per_rank(world, rows, root, mode, tmp_path):
    """The same over REAL RCCL (the library's own dlopen of librccl.so.1, the hand-declared ABI, ncclCommInitRank with the id from a file)."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs (RCCL refuses two ranks on one device); this box has %d" % (world, torch.cuda.device_count()))
    _ranks(world, rows, root, mode, tmp_path, list(range(world)), {})


def _bench2(extra, env_extra, rows="1000001"):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rows", rows, "--backend", "gloo", "--same-device", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--gather", "abi"], ["--no-codec"], ["--gather", "abi", "--root-share", "0.5"], ["--gather", "abi", "--config", "cfg4"]])
def test_bench_ships_the_shards_through_the_c_abi_gather(extra):
    """bench.py --gather abi (the default whenever the column travels as raw f64): ShardGatherer's "abi" transport end to end on two
    ranks -- the id broadcast through torch.distributed, the communicator's own rank count in the line, the gathered column verified."""
    d = _bench2(extra, {"STRSIM_RCCL_LIB": _fake_rccl()})
    dd = d["config"]["distributed"]
    assert dd["gather_impl"].startswith("abi: strsim_gather_f64_ranges") and dd["rccl_comm_ranks"] == 2
    assert d["config"]["gather_transport"].startswith("f64 (strsim_gather_f64_ranges")
    assert d["config"]["gather_verified"] is True and d["config"]["call_mode"].startswith("stream_ordered")
    assert "abi_gather_leg" not in d["config"]  # (the headline already is that gather)


@pytest.mark.gpu
def test_root_share_with_the_f64_transport_of_torch_distributed():
    """--root-share 0.5 --no-codec over torch.distributed.gather used to raise inside ShardGatherer.submit."""
    d = _bench2(["--gather", "torch", "--no-codec", "--root-share", "0.5"], {}, rows="1000000")
    assert d["config"]["rows_rank0"] == 249984 and d["config"]["gather_transport"] == "f64"
    assert d["config"]["distributed"]["gather_impl"].startswith("torch.distributed.gather") and d["config"]["gather_verified"] is True


@pytest.mark.gpu
def test_default_multi_rank_run_reports_a_c_abi_gather_leg():
    """The driver's scaling run passes no flags: the headline ships codes through torch.distributed, and the SAME run then repeats the
    steps with the C ABI's f64 gather and reports it beside the headline (config.abi_gather_leg) -- verified, with RCCL's rank count."""
    d = _bench2([], {"STRSIM_RCCL_LIB": _fake_rccl()})
    assert d["config"]["distributed"]["gather_impl"].startswith("torch.distributed.gather") and d["config"]["gather_verified"] is True
    leg = d["config"]["abi_gather_leg"]
    assert leg["ok"] is True and leg["gather_verified"] is True and leg["rccl_comm_ranks"] == 2 and leg["value"] > 0
    # without a reachable RCCL (two ranks on one GPU, no stand-in) the leg is not attempted, and the line says nothing about it
    d = _bench2([], {})
    assert "abi_gather_leg" not in d["config"] and d["config"]["gather_verified"] is True


@pytest.mark.gpu
def test_a_c_abi_gather_leg_that_cannot_start_leaves_the_headline_intact():
    """STRSIM_RCCL_LIB names a file that is not there: every rank agrees the leg cannot run, the headline line is whole."""
    d = _bench2([], {"STRSIM_RCCL_LIB": "/nonexistent/librccl.so"})
    leg = d["config"]["abi_gather_leg"]
    assert leg["ok"] is False and "error" in leg and d["config"]["gather_verified"] is True and d["value"] > 0
