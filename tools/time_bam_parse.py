#!/usr/bin/env python3
"""Throughput of the GPU BAM record splitter + field extraction on one resident slab (record bytes already in HBM).
usage: time_bam_parse.py [reads]   (rocprofv3 --kernel-trace --stats around it gives the kernels' standalone times)"""
import ctypes as C
import os
import struct
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd  # noqa: E402
import exon_amd._lib as L  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_500_000
path = f"/tmp/bamparse.{n}.ubam"
if not os.path.exists(path):
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "gen_text"), "bam", str(n), path])
raw = open(path, "rb").read()
# header: magic, l_text, text, n_ref, then per reference l_name, name, l_ref
assert raw[:4] == b"BAM\x01"
o = 8 + struct.unpack_from("<i", raw, 4)[0]
n_ref = struct.unpack_from("<i", raw, o)[0]
o += 4
for _ in range(n_ref):
    o += 8 + struct.unpack_from("<i", raw, o)[0]
body = np.frombuffer(raw, np.uint8)[o:]
ctx = exon_amd.Context(0)
p = exon_amd.BAMParser(ctx, n_ref, max_slab_bytes=len(body) + 4096)
d = ctx.to_device(np.concatenate([body, np.zeros(64, np.uint8)]))
cols = L.BAMColumns()
for rep in range(4):
    t = time.perf_counter()
    ctx._check(ctx.lib.exon_hip_bam_parser_parse(p.h, None, d.ptr, len(body), C.byref(cols)))
    ctx.sync()
    dt = time.perf_counter() - t
    print(f"rep {rep}: {cols.n_rows} reads, {len(body) / 1e6:.0f} MB in {dt * 1e3:.2f} ms = {len(body) / dt / 1e9:.1f} GB/s, {cols.n_rows / dt / 1e6:.0f} Mreads/s, undecided {cols.n_undecided}")
