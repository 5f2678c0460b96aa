#!/usr/bin/env python3
"""Throughput of the GPU VCF parser on one resident slab (text already in HBM)."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
path = "/tmp/parse_bench.vcf"
subprocess.check_call([os.path.join(ROOT, "tools", "bin", "gen_text"), "vcf", str(n), path])
raw = open(path, "rb").read()
body = raw[raw.index(b"\n1\t") + 1:]
ctx = exon_amd.Context(0)
p = exon_amd.VCFParser(ctx, ["1"], info_field="AF", max_slab_bytes=len(body) + 64)
d = ctx.to_device(np.concatenate([np.frombuffer(body, np.uint8), np.zeros(64, np.uint8)]))
for rep in range(4):
    t = time.perf_counter()
    cols = p.parse_device(d, len(body))
    dt = time.perf_counter() - t
    print(f"rep {rep}: {cols.n_rows} rows, {len(body) / 1e6:.0f} MB in {dt * 1e3:.2f} ms = {len(body) / dt / 1e9:.1f} GB/s text, {cols.n_rows / dt / 1e6:.0f} Mrows/s, undecided {cols.n_undecided}")
