#!/usr/bin/env python3
"""Timeline of one file pipeline from a rocprofv3 --kernel-trace CSV: for the LAST scan in the trace, every inflate launch with the
idle time of the inflate stream in front of it, and what the chip ran meanwhile.  usage: timeline_gaps.py kernel_trace.csv"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    short = re.split(r"[<(]", name.replace("void ", "").replace("(anonymous namespace)::", ""))[0].split("::")[-1]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
infl = [i for i, r in enumerate(rows) if r[2].startswith("k_inflate")]
if not infl:
    sys.exit("no inflate launches in the trace")
# the last scan: inflate launches separated by less than 20 ms
last = [infl[-1]]
for i in reversed(infl[:-1]):
    if rows[last[0]][0] - rows[i][1] > 20_000_000:
        break
    last.insert(0, i)
t0 = rows[last[0]][0]
end_scan = max(r[1] for r in rows[last[0]:])
print(f"last scan: {len(last)} inflate launches, first start -> last kernel end {(end_scan - t0) / 1e6:.2f} ms")
prev_end = None
busy = 0
for i in last:
    s, e, n, q = rows[i]
    crc = next((r for r in rows[i:] if r[2] == "k_crc32" and r[0] >= s), None)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    others = [(r[2], (r[0] - t0) / 1e6, (r[1] - t0) / 1e6) for r in rows if r[0] < e and r[1] > s and not r[2].startswith("k_inflate") and r[2] != "k_crc32" and r[1] - r[0] > 100_000]
    print(f"  inflate {(s - t0) / 1e6:7.2f} .. {(e - t0) / 1e6:7.2f} ms ({(e - s) / 1e6:.2f})  idle before {gap:7.1f} us  crc {((crc[1] - crc[0]) / 1e6 if crc else 0):.2f} ms | "
          + ", ".join(f"{o[0]} {o[1]:.2f}..{o[2]:.2f}" for o in others[:6]))
    busy += e - s + ((crc[1] - crc[0]) if crc else 0)
    prev_end = crc[1] if crc else e
print(f"inflate stream busy {busy / 1e6:.2f} ms of {(end_scan - t0) / 1e6:.2f} ms")
