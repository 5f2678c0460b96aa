"""K5 at the read lengths sequencers really produce (uniform L; one Arrow batch of ~1.9 GB): which path serves them and how
fast.  Path A needs L % 4 == 0; everything else (101, 151, 250 ...) takes path B."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import exon_amd
ctx = exon_amd.Context(0)
for L in [int(a) for a in sys.argv[1:]] or [100, 148, 64, 36, 101, 151, 150, 51, 76, 250, 301]:
    n = int(1.9e9) // L
    off, data = ctx.gen_c5(5, 0, n, L)
    d = ctx.zeros(np.int64, L * 256)
    best = 1e9
    for rep in range(6):
        d.zero(); ctx.sync()
        ctx.timer_start(); ctx.qual_pos_hist(off, data, n, L, d); ms = ctx.timer_stop_ms(); ctx.sync()
        if rep: best = min(best, ms)
    ok = int(d.to_host().sum()) == n * L
    print(f"L={L:4d} reads={n:9d}  {best:7.3f} ms  {n * (L + 4) / best / 1e6:6.0f} GB/s  {n * (L + 4) / best / 8e9:5.3f} of peak  sum {'ok' if ok else 'WRONG'}", flush=True)
    del off, data, d
