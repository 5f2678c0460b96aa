#!/usr/bin/env python3
"""K6 (interval overlap count) at full size: rows x 20.375 B against the HBM roofline.  Columns are generated on the
device with torch (uniform reference ids / starts, interval lengths < 20 kb)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import exon_amd
from exon_amd.engine import _col
import ctypes as C
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = exon_amd.Context(0)
g = torch.Generator(device="cuda").manual_seed(6)
ref = torch.randint(0, 25, (n,), dtype=torch.int32, device="cuda", generator=g)
start = torch.randint(1, 250_000_000, (n,), dtype=torch.int64, device="cuda", generator=g)
end = start + torch.randint(0, 20_000, (n,), dtype=torch.int64, device="cuda", generator=g)
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
c0, c1, c2 = _col(ref.data_ptr(), None, None, n), _col(start.data_ptr(), None, None, n), _col(end.data_ptr(), None, None, n)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    def launch():
        ctx._check(ctx.lib.exon_hip_overlap_count(ctx.h, s.cuda_stream, C.byref(c0), C.byref(c1), C.byref(c2), n, 6, 50_000_000, 100_000_000, cnt.data_ptr()))
    for _ in range(3):
        launch()
    s.synchronize()
    cnt.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(s); launch(); b.record(s)
    s.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in ev])
want = int(((ref == 6) & (start <= 100_000_000) & (end >= 50_000_000)).sum().item())
print(f"K6 {n} rows: {ms.mean():.3f} ms/launch (min {ms.min():.3f}) -> {n * 20.375 / ms.mean() / 1e6:.0f} GB/s = {n * 20.375 / ms.mean() / 1e6 / 8000:.1%} of 8 TB/s; "
      f"{n / ms.mean() / 1e3:.0f} Mrows/s; count {cnt.item() // 10} (torch check {want})")
