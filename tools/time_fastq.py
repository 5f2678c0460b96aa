#!/usr/bin/env python3
"""FASTQ per-position quality histogram end to end (file -> HBM -> record split -> K5 over views) vs the host decode
path, plus the resident-slab rate of the split + histogram kernels."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ragged = sys.argv[2] if len(sys.argv) > 2 else "0"
path = "/tmp/fq_bench.fastq"
subprocess.check_call([os.path.join(ROOT, "tools", "bin", "gen_text"), "fastq", str(n), path, "150", ragged])
size = os.path.getsize(path)
open(path, "rb").read()  # page cache
ctx = exon_amd.Context(0)


def run(gpu_parse):
    scan = exon_amd.Scan(path, "fastq", gpu_parse=gpu_parse)
    plan = ctx.plan_qual_pos_hist(256, columns=(3,))
    st = plan.open()
    t = time.perf_counter()
    rows = st.consume(scan)
    counts, _ = st.finish()
    dt = time.perf_counter() - t
    st.close(); plan.close(); scan.close()
    return rows, np.array(counts), dt


for rep in range(3):
    rg, hg, tg = run(True)
    print(f"gpu split: {rg} reads, {size / 1e9:.2f} GB in {tg:.3f} s = {size / tg / 1e9:.2f} GB/s file, {rg / tg / 1e6:.1f} Mreads/s")
rh, hh, th = run(False)
print(f"host decode: {rh} reads in {th:.3f} s = {size / th / 1e9:.2f} GB/s file, {rh / th / 1e6:.1f} Mreads/s; equal: {np.array_equal(hg, hh)}")

# resident slab
m = min(size, 1 << 30)
raw = np.fromfile(path, np.uint8, m)
cut = int(np.flatnonzero(raw[: m] == 10)[-1]) + 1
d = ctx.to_device(np.concatenate([raw[:cut], np.zeros(64, np.uint8)]))
p = exon_amd.FASTQParser(ctx, max_slab_bytes=cut + 64)
d_hist = ctx.zeros(np.int64, 256 * 256)
for rep in range(4):
    t = time.perf_counter()
    v = p.parse_device(d, cut, final=False)
    t1 = time.perf_counter()
    ctx.qual_pos_hist_views(d, v.qual_start, v.qual_end, v.n_reads, 256, d_hist)
    ctx.sync()
    t2 = time.perf_counter()
    print(f"resident: split {cut / (t1 - t) / 1e9:.0f} GB/s text ({(t1 - t) * 1e3:.2f} ms), hist {v.n_reads * 150 / (t2 - t1) / 1e9:.0f} GB/s quality bytes "
          f"({(t2 - t1) * 1e3:.2f} ms), total {v.n_reads / (t2 - t) / 1e6:.0f} Mreads/s, undecided {v.n_undecided}")
