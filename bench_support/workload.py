"""Synthetic workloads of BASELINE.json `configs`, generated on the GPU (or host) by libstrsim_synth.so."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
UNIFORM, ZIPF = 0, 1

# name -> (measure, rows, law, lo, hi, seed)   
CONFIGS = {
    "cfg1": ("levenshtein", 1_000_000, UNIFORM, 0, 16, 1),
    "cfg2": ("levenshtein", 100_000_000, UNIFORM, 1, 32, 2),
    "cfg3": ("jaro_winkler", 100_000_000, ZIPF, 4, 128, 3),
    "cfg4": ("all", 200_000_000, UNIFORM, 1, 32, 4),
    "cfg5": ("levenshtein", 10_000_000, UNIFORM, 1, 1024, 5),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libstrsim_synth.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", HERE, "-s"])
        L = C.CDLL(so)
        u64, u32, vp, i32 = C.c_uint64, C.c_uint32, C.c_void_p, C.c_int
        L.synth_lengths_device.argtypes = [u64, i32, u32, u32, u64, u64, vp, vp, vp]
        L.synth_fill_device.argtypes = [u64, i32, u32, u32, u64, u64, vp, vp, vp, vp, vp]
        L.synth_lengths_host.argtypes = [u64, i32, u32, u32, u64, u64, vp, vp]
        L.synth_fill_host.argtypes = [u64, i32, u32, u32, u64, u64, vp, vp, vp, vp]
        for f in (L.synth_lengths_device, L.synth_fill_device, L.synth_lengths_host, L.synth_fill_host):
            f.restype = i32
        _lib = L
    return _lib


def host_columns(seed, law, lo, hi, row0, n):
    """-> (offA u32[n+1], valA u8, offB, valB) numpy, rows [row0, row0+n) of the synthetic frame."""
    la = np.empty(n, dtype=np.uint32)
    lb = np.empty(n, dtype=np.uint32)
    assert lib().synth_lengths_host(seed, law, lo, hi, row0, n, la.ctypes.data, lb.ctypes.data) == 0
    oa = np.zeros(n + 1, dtype=np.uint64)
    ob = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(la, out=oa[1:])
    np.cumsum(lb, out=ob[1:])
    assert oa[-1] < 2**32 and ob[-1] < 2**32
    oa32, ob32 = oa.astype(np.uint32), ob.astype(np.uint32)
    va = np.empty(max(int(oa[-1]), 1), dtype=np.uint8)
    vb = np.empty(max(int(ob[-1]), 1), dtype=np.uint8)
    assert lib().synth_fill_host(seed, law, lo, hi, row0, n, oa32.ctypes.data, va.ctypes.data, ob32.ctypes.data,
                                 vb.ctypes.data) == 0
    return oa32, va[: int(oa[-1])], ob32, vb[: int(ob[-1])]


def device_columns(seed, law, lo, hi, row0, n, device):
    """Same frame rows generated on `device`; -> torch tensors (offA i32[n+1] holding u32 bits, valA u8, offB, valB)
    plus the realised byte totals."""
    import torch
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream().cuda_stream
        la = torch.empty(n, dtype=torch.int32, device=device)
        lb = torch.empty(n, dtype=torch.int32, device=device)
        rc = lib().synth_lengths_device(seed, law, lo, hi, row0, n, la.data_ptr(), lb.data_ptr(), stream)
        assert rc == 0, rc
        offs, tot = [], []
        for ln in (la, lb):
            o = torch.zeros(n + 1, dtype=torch.int64, device=device)
            torch.cumsum(ln, 0, out=o[1:])
            total = int(o[-1].item())
            assert total < 2**32, "shard values exceed the u32 offset range; split the shard"
            offs.append(o.to(torch.int32))  # wraps mod 2^32 = the u32 bit pattern
            tot.append(total)
            del o
        del la, lb
        va = torch.empty(max(tot[0], 1) + 64, dtype=torch.uint8, device=device)
        vb = torch.empty(max(tot[1], 1) + 64, dtype=torch.uint8, device=device)
        rc = lib().synth_fill_device(seed, law, lo, hi, row0, n, offs[0].data_ptr(), va.data_ptr(), offs[1].data_ptr(),
                                     vb.data_ptr(), stream)
        assert rc == 0, rc
        torch.cuda.synchronize(device)
    return offs[0], va, offs[1], vb, tot[0], tot[1]
