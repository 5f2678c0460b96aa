#!/usr/bin/env python3
"""Last N kernels of a rocprofv3 --kernel-trace csv as a timeline: start offset, duration, grid, name.  usage: kernel_timeline.py CSV [N]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    print("%10.1f us  +%9.1f us  grid %9s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:70]))
