#!/usr/bin/env python3
"""How many members of the serial inflate kernel are resident at once: time of ONE launch over the first K members of a VCF-text
slab, K around the machine-full.  The time steps up where a second round of workgroups begins.  usage: inflate_knee.py [kind] [n1,n2,...]
(small counts + EXON_HIP_INFLATE_PAR=0 / 1: where the lane-parallel decoder stops paying)"""
import ctypes as C
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import exon_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")
kind = sys.argv[1] if len(sys.argv) > 1 else "vcf"
plain, comp = f"/tmp/knee.{kind}", f"/tmp/knee.{kind}.gz"
subprocess.check_call([os.path.join(BIN, "gen_text"), kind, "14000000" if kind == "vcf" else "3000000", plain])
subprocess.check_call([os.path.join(BIN, "bgzip"), plain, comp, "6"])
raw = open(comp, "rb").read()
ctx = exon_amd.Context(0)
blocks, n, consumed, out_bytes = exon_amd.bgzf_scan(raw)
buf = np.frombuffer(raw, np.uint8)[:consumed]
d_comp = ctx.to_device(np.concatenate([buf, np.zeros(4096 + (-len(buf)) % 4, np.uint8)]))
d_out = ctx.empty(np.uint8, out_bytes + 64)
bad = C.c_int32(-1)
print(f"{kind}: {n} members available")
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (4096, 5632, 5888, 6016, 6144, 6272, 6400, 6656, 6912, 7168, 7680, 8192)
for nb in sizes:
    if nb > n:
        break
    ts = []
    for rep in range(4):
        t = time.perf_counter()
        ctx._check(ctx.lib.exon_hip_bgzf_inflate(ctx.h, None, d_comp.ptr, blocks, nb, d_out.ptr, 0, C.byref(bad)))
        ts.append(time.perf_counter() - t)
    print(f"  {nb:5d} members ({nb / 256:.1f} per CU): {min(ts) * 1e3:.2f} ms", flush=True)
