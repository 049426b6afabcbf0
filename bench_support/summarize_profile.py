#!/usr/bin/env python3
"""Condense a bench_support/profile.sh output directory into a small text summary (for profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:70]


def traffic_json(out, key, kernel="k_lane_stage"):
    """HBM bytes per launch of the dominant kernel (name contains `kernel`): FETCH_SIZE (KiB, doubled: gfx950 reports
    half of a wide coalesced read -- MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KiB), each from its own --pmc pass.
    key = "<config>:<measure>:<rows per GPU>", the key bench.py looks up."""
    import json
    vals = {}
    for grp, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(out, grp), "*counter_collection.csv")
        if not f:
            return
        # "k_a+k_b": the pass consists of both kernels -- per-launch means, added up
        tot = 0.0
        for kname in kernel.split("+"):
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                 if kname in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            if not v:
                return
            tot += sum(v) / len(v)
        vals[ctr] = tot
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    lib = os.environ.get("STRSIM_AMD_LIB") or os.path.join(root, "polars-strsim_amd", "polars_strsim", "libpolars_strsim_amd.so")
    rec = {"fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
           "traffic_bytes_per_launch": int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
           "kernel": kernel, "source": os.path.basename(out),
           # bench.py prints the figure only for the build it was measured on
           "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]}
    json.dump({key: rec}, open(os.path.join(out, "traffic_record.json"), "w"), indent=1)  # travels back in gpurun_out/
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
    try:
        db = json.load(open(path))
    except Exception:
        db = {}
    db[key] = rec
    json.dump(db, open(path, "w"), indent=1, sort_keys=True)
    print("traffic.json[%s] = %s" % (key, rec))


def main():
    out = sys.argv[1]
    if len(sys.argv) > 2:
        traffic_json(out, sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "k_lane_stage")
    st = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    if st:
        print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
        for r in csv.DictReader(open(st)):
            print(f'{short(r["Name"]):70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:10.1f} '
                  f'min_us={float(r["MinNs"])/1e3:10.1f} max_us={float(r["MaxNs"])/1e3:10.1f} pct={r["Percentage"]}')
    for grp in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_ta"):
        f = find(os.path.join(out, grp), "*counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"== {grp} (per-dispatch mean) ==")
        for k, cs in acc.items():
            if "strsim" not in k:
                continue
            print(" ", k)
            for c, v in sorted(cs.items()):
                print(f"     {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")


if __name__ == "__main__":
    main()
