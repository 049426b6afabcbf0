#!/usr/bin/env python3
"""Print the strsim kernels of a few consecutive steps from a rocprofv3 kernel_trace.csv (start/end in ns from the first)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "strsim" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
mid = rows[len(rows) // 2: len(rows) // 2 + n]
t0 = int(mid[0]["Start_Timestamp"])
for r in mid:
    print("%-28s start %7d end %7d dur %6d grid %s wg %s" % (
        r["Kernel_Name"].split("(")[0][-28:], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0,
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X") or r.get("Grid_Size"),
        r.get("Workgroup_Size_X") or r.get("Workgroup_Size")))
