#!/usr/bin/env python3
"""Timeline of the LAST plugin call in a rocprofv3 --kernel-trace --memory-copy-trace run of bench_plugin_e2e.py:
per slice the H2D copies, kernels and D2H copy with start/end relative to the call's first copy, the achieved GB/s of every
copy and the idle gaps of the H2D direction.   usage: e2e_timeline.py <rocprof output dir>"""
import csv, glob, os, sys
d = sys.argv[1]
def find(p):
    r = glob.glob(os.path.join(d, "**", p), recursive=True)
    return r[0] if r else None
cp = list(csv.DictReader(open(find("*memory_copy_trace.csv"))))
kn = list(csv.DictReader(open(find("*kernel_trace.csv"))))
ev = []
for r in cp:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), int(r.get("Bytes", r.get("Size", 0)) or 0)))
for r in kn:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:" + r["Kernel_Name"].split("(")[0][-28:], 0))
ev.sort()
# the last call = the events after the last gap of more than 3 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[:i][-8:]) > 3_000_000: cut = i
call = ev[cut:]
t0 = call[0][0]
h2d_busy = 0; last_h2d_end = None; gaps = 0
for s, e, what, b in call:
    if what.startswith("K:") and e - s < 20000: continue
    line = "%9.3f -> %9.3f ms  %-34s" % ((s - t0) / 1e6, (e - t0) / 1e6, what)
    if b: line += " %8.2f MB  %6.1f GB/s" % (b / 1e6, b / max(e - s, 1))
    print(line)
    if "HOST_TO_DEVICE" in what:
        h2d_busy += e - s
        if last_h2d_end is not None and s > last_h2d_end: gaps += s - last_h2d_end
        last_h2d_end = e
print("call span %.3f ms; H2D busy %.3f ms; gaps between H2D copies %.3f ms" % ((max(e for _, e, _, _ in call) - t0) / 1e6, h2d_busy / 1e6, gaps / 1e6))
