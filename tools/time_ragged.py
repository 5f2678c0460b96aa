import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import exon_amd
from oracle import Oracle
ctx = exon_amd.Context(0)
rng = np.random.default_rng(1)
n = 16_000_000
lens = rng.integers(50, 152, n).astype(np.int64)
off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum(lens); assert off[-1] < 2**31
off = off.astype(np.int32)
doff = ctx.to_device(off)
# device bytes: reuse the uniform generator output as raw bytes (values 33..74)
_, data = ctx.gen_c5(5, 0, (int(off[-1]) + 99) // 100, 100)
d = ctx.zeros(np.int64, 151 * 256)
for rep in range(4):
    d.zero(); ctx.sync()
    ctx.timer_start(); ctx.qual_pos_hist(doff, data, n, 151, d); ms = ctx.timer_stop_ms(); ctx.sync()
    print(f"ragged: {ms:.3f} ms  {(off[-1] + 4 * n) / ms / 1e6:.0f} GB/s")
h = d.to_host().reshape(151, 256)
print("sum check", h.sum() == off[-1])
m = 300_000
hd = data.to_host(int(off[m]))
want, _ = Oracle().c5_qual_pos_hist(off[:m + 1], hd, 151)
d2 = ctx.zeros(np.int64, 151 * 256); ctx.qual_pos_hist(doff, data, m, 151, d2); ctx.sync()
print("parity", np.array_equal(d2.to_host().reshape(151, 256), want))
