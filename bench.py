#!/usr/bin/env python3
"""bench.py -- headline benchmark: M string-pairs/s + achieved HBM GB/s (BASELINE.json `metric`).

One step = one pass of the hot path (lane-per-pair kernel + wave-per-pair kernel) over one rank's
shard of a synthetic two-column Utf8 frame that is already resident in HBM.  Default workload =
BASELINE.json configs[1]: Levenshtein, 100 M rows, <= 32-byte strings, one MI355X.  With N > 1 ranks
(launched by torch.distributed.run, one process per GPU) the SAME frame is cut by the reference's row partition
(split_offsets(rows, N), strsim.rs:21-39: rows / N each, remainder to the last rank) -- strong scaling, the metric
BASELINE.json names ("100 M rows @ 1/2/4/8 GPU") -- and each step's f64 shard is gathered to rank 0 over RCCL/xGMI on a
side stream, overlapped with the next step's kernels.  --scaling weak holds the config's row count per GPU instead.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      -- dominant kernel (k_lane_stage) vs the HBM roof, from hipEvents on its own stream
  cpu_baseline  -- the CPU oracle (C restatement of the reference, "port") on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "polars-strsim_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # defaults: the GPU needs ~20 steps (40 ms) of this load before step times settle -- with --warmup 2 --steps 10 the
    # headline kernel measures 1.89-1.93 ms, from --warmup 10 on 1.71-1.77 ms (same box, back to back)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--preheat-ms", type=float, default=40.0,
                   help="untimed steps run BEFORE the --warmup steps until the GPU has been under this load for so long "
                        "(clocks settle ~25-40 ms after idle; reported as config.preheat_steps; 0 = none)")
    p.add_argument("--config", default="cfg2", help="cfg1|cfg2|cfg3|cfg5 (BASELINE.json configs; cfg2 = headline)")
    p.add_argument("--rows", type=int, default=0, help="rows of the frame (strong) / per GPU (weak); default: the config's row count")
    p.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                   help="N>1: strong = the config's frame split over the ranks (BASELINE metric); weak = the config's row count per rank")
    p.add_argument("--force-codec", action="store_true", help="testing only: code the gathered column even when strings may exceed 32 bytes")
    p.add_argument("--measure", default="", help="override the config's measure")
    p.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather of the result shards")
    p.add_argument("--no-codec", action="store_true", help="N>1: gather raw f64 instead of 16-bit codes")
    p.add_argument("--gather", default="auto", choices=["auto", "abi", "torch"],
                   help="N>1, which gather carries the result shards: abi = the C ABI's own (strsim_gather_f64_ranges: grouped RCCL send/recv "
                        "of the raw f64 shards, north_star's literal form; implies --no-codec), torch = torch.distributed.gather (coded "
                        "when the strings allow).  auto: abi when the column travels as f64 anyway (--no-codec, cfg3, cfg5), else torch")
    p.add_argument("--no-abi-leg", action="store_true",
                   help="N>1: skip the second, separately reported leg that repeats the steps with the C ABI's f64 gather when the "
                        "headline used torch.distributed's (config.abi_gather_leg)")
    p.add_argument("--abi-leg-timeout", type=float, default=120.0, help="seconds before the C-ABI-gather leg is given up (the headline line is printed regardless)")
    p.add_argument("--no-extra-modes", action="store_true", help="skip value_default_mode and cold_first_call_ms (extra fields)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to smoke-test the control flow)")
    p.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0")
    p.add_argument("--cpu-sample-rows", type=int, default=0)
    p.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive plugin-ABI measurement (N=1 extra field)")
    p.add_argument("--root-share", type=float, default=1.0,
                   help="N>1, strong scaling: rank 0 takes this fraction of an equal share and the other ranks split the rest "
                        "(rank 0 also decodes the gathered column: profiles/r5_root_rehearsal.txt).  A DEVIATION from the reference's "
                        "partition (split_offsets, strsim.rs:21-39), off by default (1.0) and named in the JSON line when used")
    return p.parse_args()


def cpu_baseline(measure, cfg, rows_total):
    """Oracle ("port" of the reference CPU path) on a bounded prefix of the same frame, all host cores."""
    import numpy as np
    import oracle_lib as O
    from bench_support import workload as W
    _, _, law, lo, hi, seed = cfg
    logical = os.cpu_count() or 1
    cores, quota_note = logical, ""
    try:  # a cgroup CPU quota (the GPU boxes: 16 CPUs' worth of 256 logical) is what this process can really use:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # more threads than that only get throttled
        if q != "max":
            cores = max(1, min(logical, int(round(int(q) / int(per)))))
            quota_note = f", cgroup CPU quota {int(q) / int(per):g} of {logical} logical CPUs"
    except Exception:
        pass
    n = 250_000 * max(1, min(cores, 64) // 4)
    best = None
    while True:
        n = min(n, rows_total)
        oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, n)
        t0 = time.perf_counter()
        out = O.batch(measure, oa, va, ob, vb, nthreads=cores)
        dt = time.perf_counter() - t0
        best = (n, dt, out)
        if dt >= 8.0 or n >= rows_total or n >= 64_000_000:
            break
        n = int(n * max(2.0, min(8.0, 12.0 / max(dt, 1e-3))))
    n, dt, out = best
    return {"value": n / dt / 1e6, "unit": "M string-pairs/s", "cores": cores, "kind": "port",
            "sample": f"first {n} rows of the same synthetic frame, {dt:.1f} s wall, oracle/strsim_oracle.c on {cores} threads "
                      f"(split_offsets partition){quota_note}"}, (n, out)


def plugin_e2e(measure, cfg, rows, name=None):
    """PCIe-inclusive rate through _polars_plugin_<measure> (host Arrow string views in, host f64 out) on a prefix.
    Measured in a CHILD process (this file, --e2e-child): what bounds such a call is host CPU time against the container's
    quota, and this process still carries torch's and the generator's thread pools (in-process the same call measured
    30-40 % slower than in a fresh interpreter)."""
    if name is not None:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--e2e-child", name, measure, str(rows)],
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and line:
                return json.loads(line[-1])
        except Exception:
            pass  # fall through: measure here
    import numpy as np
    import pyarrow as pa
    from bench_support import workload as W
    from strsim_amd import arrow_host as H
    _, _, law, lo, hi, seed = cfg
    n = min(rows, 8_000_000 if hi <= 128 else 200_000)
    oa, va, ob, vb = W.host_columns(seed, law, lo, hi, 0, n)
    mk = lambda o, v: pa.StringArray.from_buffers(n, pa.py_buffer(o.astype(np.int32)), pa.py_buffer(v)).cast(pa.string_view())
    a, b = mk(oa, va), mk(ob, vb)
    H.call_plugin(measure, a[:1000], b[:1000])
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        H.call_plugin(measure, a, b)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": n / best / 1e6, "unit": "M string-pairs/s", "rows": n, "layout": "Utf8View",
            "note": "host Arrow views -> host f64 through the plugin C ABI (pack + H2D + kernels + D2H); never `value`"}


def self_launch(a):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks HERE, one process per GPU,
    as a CHILD `python -m torch.distributed.run` (never an exec of this process; nothing here has touched the GPU yet --
    torch.cuda.device_count() does not initialise it), relay what the ranks print (rank 0's JSON line) and leave with the child's
    exit code.  N ranks on fewer than N GPUs are refused here already (the ranks check it again among themselves)."""
    import socket
    import subprocess
    if not a.same_device:
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but this node has {have} GPU(s) (one rank per GPU; --same-device --backend gloo is the one-GPU smoke test)")
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no launcher in the environment: starting %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    import torch
    import torch.distributed as dist
    from bench_support import workload as W
    import strsim_amd as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:  # (WORLD_SIZE=1 set by hand; without WORLD_SIZE the ranks were started above)
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE=1: unset it (bench.py then starts the ranks itself) or launch {a.gpus} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if a.same_device:
        # every rank on cuda:0 is a smoke test of the control flow, never a scaling measurement: RCCL refuses two ranks on one
        # device, so it only exists with the gloo backend -- and the JSON line says so (config.distributed.same_device)
        if a.backend == "nccl":
            raise SystemExit("--same-device needs --backend gloo (RCCL wants one GPU per rank)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    # who is really taking part: the world torch.distributed has formed, its backend, every rank's device (a scaling record must
    # be readable without trusting the command line)
    distributed = {"world_size": 1, "backend": None, "same_device": bool(a.same_device), "devices": None}
    if world > 1:
        assert dist.get_world_size() == world == a.gpus, (dist.get_world_size(), world, a.gpus)
        cdev0 = dev if a.backend == "nccl" else "cpu"
        props = torch.cuda.get_device_properties(dev)
        ident = torch.tensor([rank, local_rank, torch.cuda.current_device(), int(getattr(props, "pci_bus_id", -1)),
                              int(getattr(props, "pci_device_id", -1)), props.multi_processor_count], dtype=torch.int64, device=cdev0)
        idents = [torch.zeros_like(ident) for _ in range(world)]
        dist.all_gather(idents, ident)
        devices = [{"rank": int(v[0]), "local_rank": int(v[1]), "cuda_device": int(v[2]), "pci_bus": int(v[3]), "pci_device": int(v[4]),
                    "compute_units": int(v[5])} for v in idents]
        distinct = len({(d["cuda_device"], d["pci_bus"], d["pci_device"]) for d in devices})
        if not a.same_device and distinct != world:
            raise SystemExit(f"{world} ranks but {distinct} distinct GPUs: {devices}")
        distributed = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "same_device": bool(a.same_device),
                       "distinct_devices": distinct, "devices": devices}

    cfg = W.CONFIGS[a.config]
    measure = a.measure or cfg[0]
    measures = list(S.MEASURES) if measure == "all" else [measure]  # cfg4: five passes over the same frame per step
    _, _, law, lo, hi, seed = cfg
    # the frame and this rank's shard of it
    if a.scaling == "weak":
        total_rows = (a.rows or cfg[1]) * world
        shards = [(r * (total_rows // world), total_rows // world) for r in range(world)]
    else:
        total_rows = a.rows or cfg[1]
        from strsim_amd.distributed import shard_ranges
        try:
            shards = shard_ranges(total_rows, world, a.root_share)  # strsim.rs:21-39 (root_share 1.0: exactly)
        except ValueError as e:
            raise SystemExit("--root-share: %s" % e)
    row_base, rows = shards[rank]

    # a shard whose packed values would not fit 32-bit offsets (cfg5: ~5 GB per column) is held as several row batches,
    # each its own offsets+values pair
    mean_len = (lo + hi) / 2.0 if law == W.UNIFORM else 26.0
    nparts = max(1, int(rows * mean_len * 1.15 / 3.5e9) + (1 if rows * mean_len * 1.15 > 3.5e9 else 0))
    bounds = [rows * p // nparts for p in range(nparts + 1)]
    parts = []
    bytesA = bytesB = 0
    for p in range(nparts):
        r0, r1 = bounds[p], bounds[p + 1]
        oa, va, ob, vb, ba, bb = W.device_columns(seed, law, lo, hi, row_base + r0, r1 - r0, dev)
        parts.append((r0, r1, oa, va, ob, vb))
        bytesA += ba
        bytesB += bb
    offA, valA, offB, valB = parts[0][2:6]
    out = [torch.empty(rows, dtype=torch.float64, device=dev) for _ in range(2 * len(measures))]

    compute_stream = torch.cuda.Stream()  # an explicit stream: handle 0 (the legacy default stream) would make the
    torch.cuda.set_stream(compute_stream)  # context create its own, and torch-side waits would not see the kernels
    ctx = S.Context(local_rank, stream=compute_stream.cuda_stream)
    assert ctx.stream == compute_stream.cuda_stream
    gather = world > 1 and not a.no_gather
    # A step's results are only read after ctx.synchronize() when nothing is gathered: one-launch calls (ABI 1.4 opt-in).  The
    # gather reads a step's results behind an event, without retiring the call first: it needs the default, stream-ordered mode.
    ctx.set_stream_ordered(bool(gather))
    shipper = None
    gather_note = None
    codec_chars = 32 if ((hi <= 32 or a.force_codec) and not a.no_codec and a.gather != "abi") else None
    rccl_reachable = a.backend == "nccl" or bool(os.environ.get("STRSIM_RCCL_LIB"))  # (real RCCL refuses two ranks on one GPU)
    impl = a.gather if a.gather != "auto" else ("abi" if codec_chars is None and rccl_reachable else "torch")

    def agree(ok):  # every rank takes the same decision: MIN over the ranks' flags (a collective every rank reaches)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def abi_gather_available():
        """Non-collective: can this rank load an RCCL and make a unique id?  Agreed on before any rank enters ncclCommInitRank."""
        from strsim_amd.distributed import AbiGather
        try:
            AbiGather.unique_id()
            return True, None
        except Exception as e:
            return False, repr(e)[:300]

    if gather:
        from strsim_amd.distributed import ShardGatherer, gather_column
        # (0) the C ABI's gather builds its communicator inside the constructor (collective): make sure every rank can get there
        if impl == "abi":
            ok, why = abi_gather_available()
            if not agree(ok):
                gather_note = "--gather abi fell back to torch.distributed.gather: " + (why or "RCCL is not loadable on another rank")
                impl = "torch"
                codec_chars = 32 if ((hi <= 32 or a.force_codec) and not a.no_codec) else None
        # (1) the fallible NON-collective setup first, agreed on before any rank enters a gather: a rank that failed here must
        #     not skip a collective the others are already inside (mismatched collectives hang on RCCL)
        ok = 1
        try:
            # strings of at most 32 characters take < 2^16 distinct similarity values: ship 16-bit codes, decode on rank 0
            shipper = ShardGatherer(ctx, compute_stream, measures, rows, dev, backend=a.backend, parts=shards,
                                    codec_chars=codec_chars, impl=impl)
            probe = torch.zeros(64, dtype=torch.uint8, device="cpu" if shipper.host else dev)
        except Exception as e:  # reported in the JSON line (gather_f64_to_rank0 false + gather_note), never silent
            ok = 0
            gather_note = "gather disabled: its setup failed: " + repr(e)[:300]
        if not agree(ok):
            gather, shipper = False, None
            gather_note = gather_note or "gather disabled: its setup failed on another rank"
        else:
            # (2) preflight, outside any timing, entered by ALL ranks together: one tiny gather over the same call path builds
            #     the communicator (so that it is not built inside the timed region when --warmup is 0) and shows whether this
            #     backend can gather at all.  (A collective that throws on some ranks only leaves no safe way on: the flags below
            #     are still exchanged, and a rank that cannot reach that exchange fails the job loudly instead of hanging it.)
            ok = 1
            try:
                gather_column(probe, world * 64, dst=0)
                torch.cuda.synchronize()
            except Exception as e:
                ok = 0
                gather_note = "gather disabled after a failed preflight: " + repr(e)[:300]
            if not agree(ok):
                gather, shipper = False, None
                gather_note = gather_note or "gather disabled: the preflight failed on another rank"
        if not gather:
            print(f"[bench rank {rank}] {gather_note}", file=sys.stderr)
            ctx.set_stream_ordered(False)  # (ShardGatherer had put the context into stream-ordered mode: nothing reads behind an event now)
    fused = len(measures) == 5  # cfg4: strsim_pairs_device_all, one fused pass with five outputs

    def step(i):
        par = i & 1
        os_ = out[par * len(measures):(par + 1) * len(measures)]
        if gather:
            for k in range(len(measures)):
                shipper.wait_slot((par, k))  # the buffers about to be overwritten must have been shipped
        for r0, r1, oa, va, ob, vb in parts:
            if fused:
                ctx.pairs_device_all(oa, va, ob, vb, outs=[o[r0:r1] for o in os_])
            else:
                for m, o in zip(measures, os_):
                    ctx.pairs_device(m, oa, va, ob, vb, out=o[r0:r1])
        if gather:
            for k, (m, o) in enumerate(zip(measures, os_)):
                shipper.submit((par, k), m, o)

    def drain():
        ctx.synchronize()
        if shipper is not None:
            shipper.drain()
        torch.cuda.synchronize()

    # One-launch calls stay pending until they are retired, and the context's ring holds 32 of them: a caller that never retires
    # makes strsim_pairs_device synchronise the stream at every wrap -- inside the timed loop.  So the loop keeps a WINDOW: calls are
    # retired in batches of about eight (one event per batch, recorded behind its last step), the oldest batch -- complete by its
    # event -- whenever another batch would take the calls in flight past 24.
    calls_per_step = len(parts) * (1 if fused else len(measures))
    batch_steps = max(1, 8 // calls_per_step)
    INFLIGHT_CALLS = max(24, 2 * calls_per_step * batch_steps)  # (< 32 for every config of BASELINE.json)
    windowed = not gather

    def run(nsteps):
        batches = []  # (event behind the batch's last step, calls in the batch), oldest first
        pending = in_batch = 0
        for i in range(nsteps):
            if windowed and in_batch == 0 and batches and pending + calls_per_step * batch_steps > INFLIGHT_CALLS:
                ev, ncalls = batches.pop(0)
                ev.synchronize()
                for _ in range(ncalls):
                    ctx.retire_oldest()
                pending -= ncalls
            step(i)
            if windowed:
                in_batch += 1
                if in_batch == batch_steps or i == nsteps - 1:
                    ev = torch.cuda.Event()
                    ev.record(compute_stream)
                    batches.append((ev, in_batch * calls_per_step))
                    pending += in_batch * calls_per_step
                    in_batch = 0
        drain()

    # Preheat (disclosed in the JSON line): the first 25-40 ms after the GPU goes from idle to this load run ~5-10 % slower
    # (clock ramp), which a short run (--warmup 5 --steps 20 is 35 ms) would measure instead of the steady state a 100 M-row
    # job is in.  One step is timed to size the preheat; every rank runs the same number of steps (the gather is collective).
    preheat_steps = 0
    if a.preheat_ms > 0:
        step(0)
        drain()
        t0 = time.perf_counter()
        step(1)
        drain()
        one = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=(dev if a.backend == "nccl" else "cpu") if world > 1 else "cpu")
        if world > 1:
            dist.all_reduce(one, op=dist.ReduceOp.MAX)
        preheat_steps = max(0, min(2000, int(a.preheat_ms * 1e-3 / max(float(one.item()), 1e-6)) + 1))
        preheat_steps += preheat_steps & 1  # (an even count: the output / gather slots alternate with the step index)
        run(preheat_steps)
        preheat_steps += 2
    run(a.warmup)
    ctx.timing(True)
    if shipper is not None:
        shipper.timing_begin()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops0 = ctx.enqueued_ops
    t0 = time.perf_counter()
    run(a.steps)
    ops_per_step = (ctx.enqueued_ops - ops0) / max(a.steps, 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = ctx.timing_read()
    ctx.timing(False)
    wave_rows = ctx.last_wave_rows
    ship_ms, decode_ms = shipper.timing_end() if shipper is not None else (None, None)
    # what each rank spent per step, so that a scaling record can be read: kernels, its shipment, its wall clock
    per_rank = None
    if world > 1:
        cdev_ = dev if a.backend == "nccl" else "cpu"
        mine_ = torch.tensor([tm["lane_ms"] / max(tm["lane_launches"], 1) * nparts,
                              tm["wave_ms"] / max(tm["wave_launches"], 1) * nparts, ship_ms if ship_ms is not None else -1.0,
                              dt / a.steps * 1e3, float(rows)], dtype=torch.float64, device=cdev_)
        all_ = [torch.zeros_like(mine_) for _ in range(world)]
        dist.all_gather(all_, mine_)
        per_rank = [{"rank": r, "rows": int(v[4].item()), "lane_kernel_ms_per_pass": round(float(v[0].item()), 4),
                     "slow_row_kernels_ms_per_pass": round(float(v[1].item()), 4),
                     "shipment_ms_per_column": None if float(v[2].item()) < 0 else round(float(v[2].item()), 4),
                     "wall_ms_per_step": round(float(v[3].item()), 4)} for r, v in enumerate(all_)]

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def verify_gather(sh, nsteps):
        """Outside any timed region: what rank 0 holds for the last shipped column must equal every rank's shard (64-bit sums of the
        bit patterns, shard by shard).  Collective; the verdict on rank 0, None elsewhere."""
        cdev = dev if a.backend == "nccl" else "cpu"
        last = out[((nsteps - 1) & 1) * len(measures) + len(measures) - 1]
        mine = last.view(torch.int64).sum().reshape(1).to(cdev)
        sums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sums, mine)
        if rank != 0:
            return None
        got = sh.result()
        return all(int(got[o:o + ln].view(torch.int64).sum().item()) == int(sums[r].item()) for r, (o, ln) in enumerate(shards))

    dt = max_over_ranks(dt)
    gather_ok = verify_gather(shipper, a.steps) if (world > 1 and gather) else None
    call_mode_timed = "stream_ordered (ABI default)" if ctx.stream_ordered else "one_launch (opt-in, strsim_ctx_set_stream_ordered(ctx, 0))"
    gather_impl = None if not gather else ("abi: strsim_gather_f64_ranges (grouped RCCL send/recv into the root's column)" if shipper.impl == "abi"
                                           else "torch.distributed.gather (shards padded to the longest)")
    rccl_comm_ranks = shipper.comm_ranks() if gather else None

    # The same steps in the ABI's DEFAULT mode (every kernel of the chain up front: what a caller who never opts in gets), timed the
    # same way right after the headline -- an extra field, never `value`.  N = 1 only: under a gather the
    # headline already runs in that mode.
    default_mode = None
    if world == 1 and not a.no_extra_modes and not ctx.stream_ordered:
        ctx.set_stream_ordered(True)
        run(min(a.warmup, 5))
        torch.cuda.synchronize()
        ops0 = ctx.enqueued_ops
        t0 = time.perf_counter()
        run(a.steps)
        torch.cuda.synchronize()
        dt_d = time.perf_counter() - t0
        default_mode = {"value": total_rows * a.steps / dt_d / 1e6, "ms_per_step": dt_d / a.steps * 1e3,
                        "enqueued_kernels_and_copies_per_step": (ctx.enqueued_ops - ops0) / max(a.steps, 1),
                        "call_mode": "stream_ordered (ABI default)", "steps": a.steps, "warmup": min(a.warmup, 5)}
        ctx.set_stream_ordered(False)

    # What ONE call costs a caller who does not loop, measured AFTER everything else so that the idle seconds do not
    # reach into the headline (two 1 s sleeps in front of the preheat cost it 0.7 %, profiles/r6_bench_lines.jsonl): (a) the first call of
    # a NEW context on a GPU that has been idle for a second -- its workspace allocations and idle clocks; the code object is already
    # loaded by then -- and (b) a call of the warm context after another idle second.  Wall clock around the call(s) of one pass over
    # this rank's shard + synchronize, no gather.
    cold_first_call_ms = idle_gpu_call_ms = None
    if not a.no_extra_modes:
        def one_pass(c):
            os_ = out[:len(measures)]
            for r0, r1, oa, va, ob, vb in parts:
                if fused:
                    c.pairs_device_all(oa, va, ob, vb, outs=[o[r0:r1] for o in os_])
                else:
                    for m, o in zip(measures, os_):
                        c.pairs_device(m, oa, va, ob, vb, out=o[r0:r1])
            c.synchronize()
        drain()
        torch.cuda.synchronize()
        time.sleep(1.0)
        t0 = time.perf_counter()
        cold_stream = torch.cuda.Stream()
        with S.Context(local_rank, stream=cold_stream.cuda_stream) as cold_ctx:
            one_pass(cold_ctx)
            cold_first_call_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        time.sleep(1.0)
        t0 = time.perf_counter()
        one_pass(ctx)
        idle_gpu_call_ms = (time.perf_counter() - t0) * 1e3

    res = None
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = total_rows * a.steps / dt / 1e6
        # each byte/offset counted once; per launch of the dominant kernel on THIS rank's shard (the
        # roofline object describes one GPU's kernel, `value` the whole job)
        read_bytes = bytesA + bytesB + 2 * 4 * (rows + nparts)
        write_bytes = 8 * rows
        lane_ms = tm["lane_ms"] / max(tm["lane_launches"], 1) * nparts   # per pass over the whole shard
        wave_ms = tm["wave_ms"] / max(tm["wave_launches"], 1) * nparts
        # HBM traffic of the dominant kernel from a separate rocprofv3 --pmc run of this same command
        # (bench_support/profile.sh -> profiles/traffic.json; FETCH_SIZE doubled per the gfx950 note)
        traffic, traffic_source = None, "none (no counter run of this library build and workload in profiles/traffic.json)"
        try:
            import hashlib
            from strsim_amd import _lib as L_
            lib_sha = hashlib.sha256(open(L_.LIB_PATH, "rb").read()).hexdigest()[:16]
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            key = f"{a.config}:{measures[0]}:{rows}"
            # only a counter run of THIS build counts (bench_support/profile.sh records the library's hash with the figure)
            if len(measures) == 1 and key in tj and tj[key].get("lib_sha256") == lib_sha:
                traffic = tj[key]["traffic_bytes_per_launch"] * nparts  # per pass over the whole shard, like `achieved`
                traffic_source = "replayed: profiles/traffic.json[%s]@lib sha256 %s (rocprofv3 --pmc passes of bench_support/profile.sh)" % (key, lib_sha)
        except Exception:
            pass
        # the dominant kernel: k_lane_stage, unless the slow-row chain after it (k_lane_wide / k_lane_utf8 /
        # k_wave_pairs, timed together by the second event pair) takes longer -- cfg3 and cfg5
        dom_ms, dom_name = lane_ms, ("k_lane_stage<%s>" if len(measures) == 1 else "k_lane_stage_all (five outputs)%s") % (measures[0] if len(measures) == 1 else "")
        if wave_ms > 0.2 * lane_ms:
            # a frame with many slow rows (cfg3, cfg5): the figure is over the WHOLE PASS -- the staged kernel reads every byte
            # of both columns, the slow-row kernels read the slow rows' bytes again; the frame's algorithmic bytes over the sum
            # of both durations, not over the longer kernel alone
            dom_ms = lane_ms + wave_ms
            dom_name = ("whole pass: k_lane_stage<%s> + " + ("k_wave_pairs" if (a.config == "cfg5" or hi > 128) else "k_lane_wide (+ k_lane_utf8, k_wave_pairs)")) % measures[0]
        achieved = read_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        res = {
            "metric": "M string-pairs/s, %s, %d M rows%s (+ achieved HBM GB/s in roofline)%s" %
                      (measure, (total_rows if a.scaling == "strong" else rows) // 1_000_000, " per GPU" if a.scaling == "weak" else "",
                       "; %d untimed preheat steps before the %d warm-up steps (--preheat-ms %g)" % (preheat_steps, a.warmup, a.preheat_ms) if preheat_steps else ""),
            "passes_per_step": len(measures),
            "value": value, "unit": "M string-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "u8/u32 bit-parallel, f64 epilogue", "data": "synthetic",
            "config": {"workload": f"{a.config}: {measure}, {total_rows} rows ({a.scaling} scaling: {rows} on rank 0), lengths "
                                   f"{'U' if law == W.UNIFORM else 'Zipf'}{{{lo}..{hi}}} bytes, a-z, seed {seed}",
                       "rows_total": total_rows, "rows_rank0": rows, "preheat_steps": preheat_steps,
                       # how the calls were enqueued: one-launch calls (strsim_ctx_set_stream_ordered(ctx, 0), the ABI's opt-in mode:
                       # a call is its first kernel alone unless the context's last call left slow rows) when nothing is gathered,
                       # the default stream-ordered mode (every kernel of the chain up front) under the gather
                       "call_mode": call_mode_timed,  # (read from the context: strsim_ctx_get_stream_ordered)
                       "calls_in_flight_max": INFLIGHT_CALLS if windowed else None,
                       "partition": "split_offsets(rows, N) (strsim.rs:21-39)" if (world == 1 or a.root_share == 1.0 or a.scaling == "weak")
                                    else "DEVIATION --root-share %g: rank 0 holds %d rows, the others split the rest by split_offsets" % (a.root_share, rows),
                       "distributed": dict(distributed, gather_impl=gather_impl, rccl_comm_ranks=rccl_comm_ranks),
                       "gather_f64_to_rank0": bool(gather),
                       "gather_transport": shipper.transport if shipper else None,
                       "codec_exceptions": shipper.exceptions() if shipper else None,
                       "gather_verified": gather_ok, "gather_note": gather_note,
                       "rows_on_wave_kernel": wave_rows, "enqueued_kernels_and_copies_per_step": ops_per_step,
                       "per_rank": per_rank, "root_decode_ms_per_column": None if decode_ms is None else round(decode_ms, 4)},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # `traffic` is never measured inside this run (PMC counters need their own rocprofv3 passes): it is the
                         # figure bench_support/profile.sh recorded for THIS library build and THIS workload, or nothing
                         "traffic_source": traffic_source,
                         "algorithmic_read_bytes": read_bytes, "algorithmic_write_bytes": write_bytes,
                         "kernel_ms": lane_ms, "wave_kernel_ms": wave_ms,
                         "achieved_read_plus_write": (read_bytes + write_bytes) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0},
        }
        res["value_default_mode"] = None if default_mode is None else round(default_mode["value"], 3)
        res["default_mode"] = default_mode if default_mode is not None else (
            "the headline ran in the default mode" if ctx.stream_ordered or gather else None)
        res["cold_first_call_ms"] = None if cold_first_call_ms is None else round(cold_first_call_ms, 3)
        res["idle_gpu_call_ms"] = None if idle_gpu_call_ms is None else round(idle_gpu_call_ms, 3)
        if a.config == "cfg5" or hi > 128:
            cells = 0.0
            for _r0, _r1, oa, _va, ob, _vb in parts:
                cells += float(((oa[1:] - oa[:-1]).to(torch.float64) * (ob[1:] - ob[:-1]).to(torch.float64)).sum().item())
            res["gcups"] = cells * len(measures) * a.steps / dt / 1e9  # DP cells per second (compute-bound workloads)
        if not a.no_e2e and world == 1 and len(measures) == 1:
            try:
                res["end_to_end_plugin_abi"] = plugin_e2e(measures[0], cfg, rows, a.config)
            except Exception as e:
                res["end_to_end_plugin_abi"] = {"value": None, "note": "failed: %r" % (e,)}
        if not a.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N=1 only
            try:
                cb, (n_s, exp) = cpu_baseline(measures[0], cfg, rows)
                res["cpu_baseline"] = cb
                got = out[((a.steps - 1) & 1) * len(measures)][:n_s].cpu().numpy()
                import numpy as np
                res["parity_vs_oracle_on_sample"] = {"rows": int(n_s), "bit_mismatches": int((got.view(np.uint64) != exp.view(np.uint64)).sum())}
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the bench line
                res["cpu_baseline"] = {"value": None, "unit": "M string-pairs/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
    # N > 1, headline gathered by torch.distributed (coded): the same steps once more with the C ABI's OWN gather (raw f64 over grouped
    # RCCL send/recv, north_star's literal form), reported beside the headline as config.abi_gather_leg -- so that the driver's
    # default scaling run is also the first contact of strsim_gather_* with real RCCL ranks.  Guarded: the headline
    # line is complete before the leg starts, and a leg that hangs (a collective some rank never reaches) is given up by a timer that
    # prints the line as it stands and leaves.
    leg_failed = False
    if world > 1 and gather and shipper.impl == "torch" and not a.no_abi_leg and rccl_reachable:
        import threading
        leg, lock, done = {"ok": False, "stage": "availability"}, threading.Lock(), [False]

        def give_up():
            with lock:
                if done[0]:
                    return
                if rank == 0:
                    res["config"]["abi_gather_leg"] = dict(leg, ok=False, error="given up after %g s (--abi-leg-timeout) in stage '%s'" % (a.abi_leg_timeout, leg["stage"]))
                    print(json.dumps(res), flush=True)
                if rank != 0:
                    time.sleep(3.0)  # (rank 0's line first: a launcher that sees a rank leave may end the others)
                os._exit(0)

        timer = threading.Timer(a.abi_leg_timeout, give_up)
        timer.daemon = True
        timer.start()
        try:
            ok, why = abi_gather_available()
            if not agree(ok):
                leg["error"] = why or "RCCL is not loadable on another rank"
            else:
                leg["stage"] = "communicator (ncclCommInitRank)"
                headline_shipper = shipper
                shipper = ShardGatherer(ctx, compute_stream, measures, rows, dev, backend=a.backend, parts=shards, codec_chars=None, impl="abi")
                leg["stage"] = "steps"
                run(min(a.warmup, 5))
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(a.steps)
                dist.barrier()
                torch.cuda.synchronize()
                dt_l = max_over_ranks(time.perf_counter() - t0)
                leg["stage"] = "verification"
                ok_l = verify_gather(shipper, a.steps)
                leg.update(ok=True, stage="done", value=total_rows * a.steps / dt_l / 1e6, unit="M string-pairs/s", ms_per_step=dt_l / a.steps * 1e3,
                           steps=a.steps, warmup=min(a.warmup, 5), gather_impl="abi: strsim_gather_f64_ranges", transport=shipper.transport,
                           rccl_comm_ranks=shipper.comm_ranks(), gather_verified=ok_l)
                shipper.close()
                shipper = headline_shipper
        except Exception as e:
            leg["error"] = repr(e)[:300]
        with lock:
            done[0] = True
        timer.cancel()
        leg_failed = not leg["ok"]
        if rank == 0:
            res["config"]["abi_gather_leg"] = leg
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        if leg_failed:  # ranks may have left the leg at different collectives: do not wait for one another
            os._exit(0)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "--e2e-child":  # bench.py --e2e-child <config> <measure> <rows>
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "polars-strsim_amd"))
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from bench_support import workload as _W
        print(json.dumps(plugin_e2e(sys.argv[3], _W.CONFIGS[sys.argv[2]], int(sys.argv[4]))), flush=True)
    else:
        main()
