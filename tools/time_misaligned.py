#!/usr/bin/env python3
"""What a column base that is 16-byte but not 256-byte aligned costs K2 / K3 / K4 (a sliced device batch).  The table is
generated once; the launches then start `skip` rows into it (skip % 16 == 0 keeps bitmaps on byte boundaries and u8 columns 16-byte aligned)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, exon_amd
ctx = exon_amd.Context(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
for kind in ("c4", "c2", "c3"):
    wl = bench.Workload(ctx, kind, rows + 4096, 0, rows + 4096)
    s = torch.cuda.current_stream().cuda_stream
    for skip in (0, 16, 32, 64, 1040):  # multiples of 16 rows: the u8 column stays 16-byte aligned
        cols = []
        for (ptr, valid, off), width in zip(wl.cols, {"c4": (4, 4, 4), "c2": (4, 8), "c3": (4, 1, 4)}[kind]):
            cols.append((ptr + skip * width, None if valid is None else valid + skip // 8, off))
        go = wl.plan.prepared(cols, rows, wl.state.data_ptr(), overwrite=True, stream=s)
        for _ in range(3): go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            e0.record()
            for _ in range(10): go()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        gbs = rows * bench.BYTES_PER_ROW[kind] / best / 1e6
        print(f"{kind} rows={rows} skip={skip:5d} rows ({skip * 4:5d} B on the f32/i32 columns): {best:.4f} ms  {gbs:.0f} GB/s  {gbs / 8000:.3f}", flush=True)
    del wl
    torch.cuda.empty_cache()
