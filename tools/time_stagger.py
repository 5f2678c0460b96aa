#!/usr/bin/env python3
"""Does it matter where the three 4 GB columns of config 4 start RELATIVE to each other?  (Allocated one after the other they all
begin on 2 MiB boundaries, so a tile's three streams hit the same offsets modulo every power of two.)  The table is generated once;
launches then start column k `skip[k]` rows into its buffer (multiples of 16 rows: bitmaps stay on byte boundaries)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, exon_amd
ctx = exon_amd.Context(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
pad = 8 * 1024 * 1024
wl = bench.Workload(ctx, "c4", rows + pad, 0, rows + pad)
s = torch.cuda.current_stream().cuda_stream
for rep_outer in range(2):
    for skips in ((0, 0, 0), (0, 1024, 2048), (0, 16, 32), (0, 64, 128), (0, 16384, 32768), (0, 262144, 524288), (0, 348160, 696320),
                  (0, 3000016, 7000032), (0, 0, 0)):
        cols = []
        for (ptr, valid, off), skip in zip(wl.cols, skips):
            cols.append((ptr + skip * 4, None if valid is None else valid + skip // 8, off))
        go = wl.plan.prepared(cols, rows, wl.state.data_ptr(), overwrite=True, stream=s)
        for _ in range(3): go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(4):
            e0.record()
            for _ in range(10): go()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        gbs = rows * bench.BYTES_PER_ROW["c4"] / best / 1e6
        print(f"c4 rows={rows} column starts +{[k * 4 for k in skips]} B: {best:.4f} ms  {gbs:.0f} GB/s  {gbs / 8000:.3f}", flush=True)
