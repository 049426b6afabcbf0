"""Does torch's first GPU init still work in a process that has already driven the GPU through libpolars_strsim_amd.so?"""
import os, sys
sys.path.insert(0, "polars-strsim_amd")
import numpy as np
import strsim_amd as S

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
strings = ["kitten", "sitting", "", "flaw"] * 100
ao, av = S.pack_strings(strings)
for i in range(n_ctx):
    with S.Context(0) as ctx:
        out = ctx.pairs_host("levenshtein", ao, av, ao, av)
print("lib calls done:", n_ctx, "contexts; out[0..3]", out[:4], flush=True)
print("env HIP_VISIBLE_DEVICES=%r ROCR_VISIBLE_DEVICES=%r" % (os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")), flush=True)
import torch
print("torch.cuda.device_count()", torch.cuda.device_count(), flush=True)
try:
    torch.cuda.init()
    x = torch.ones(4, device="cuda")
    print("torch init ok", x.sum().item(), flush=True)
except Exception as e:
    print("torch init FAILED:", e, flush=True)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
print("\n".join(libs))
