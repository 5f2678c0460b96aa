#!/usr/bin/env python3
"""Batches per second of a scan: the host reader against the GPU pipeline (exon_hip_scan_bind_ctx).  Batches are taken and released
without a pyarrow import (that would time pyarrow).  usage: time_scan_batches.py FILE {vcf|bcf|bam|sam} [info_field]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd  # noqa: E402

path, kind = sys.argv[1], sys.argv[2]
info = sys.argv[3] if len(sys.argv) > 3 else ("AF" if kind in ("vcf", "bcf") else None)
REL = C.CFUNCTYPE(None, C.c_void_p)
ctx = exon_amd.Context(0)


def drain(scan):
    rows = batches = 0
    while True:
        arr = scan.next_raw()
        if arr is None:
            return rows, batches
        rows += arr.length
        batches += 1
        REL(arr.release)(C.addressof(arr))


for label, gpu in (("host reader", False), ("GPU pipeline", True), ("GPU pipeline", True)):
    t0 = time.perf_counter()
    s = exon_amd.Scan(path, kind, info_field=info, gpu_parse=gpu)
    if gpu:
        s.bind_ctx(ctx)
    rows, batches = drain(s)
    flags = s.decoded_on_gpu() if gpu else (False, False)
    s.close()
    dt = time.perf_counter() - t0
    print(f"{kind} {label}: {rows} rows in {batches} batches, {dt * 1e3:.1f} ms = {rows / dt / 1e6:.1f} Mrows/s (decoded on the GPU: {flags[0]})", flush=True)
