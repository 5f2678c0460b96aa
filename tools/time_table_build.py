#!/usr/bin/env python3
"""Cost of one DEFLATE block header (dynamic tables) in k_inflate: same payload, k sub-blocks per BGZF block."""
import os, struct, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import exon_amd
from test_gpu_inflate import vcf_like
ctx = exon_amd.Context(0)
text = vcf_like(1200, seed=4)[:60000]


def member(data, k):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    step = (len(data) + k - 1) // k
    c = b""
    for i in range(0, len(data), step):
        c += co.compress(data[i:i + step])
        if i + step < len(data):
            c += co.flush(zlib.Z_FULL_FLUSH)
    c += co.flush()
    bsize = 18 + len(c) + 8
    assert bsize <= 65536
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + c + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


for k in (1, 4, 16, 32):
    f = member(text, k) * 6000
    best = 1e9
    for _ in range(3):
        got, dt = ctx.bgzf_inflate(f, verify_crc=False)
        best = min(best, dt)
    ok = got[:len(text)].tobytes() == text
    print(f"{k:3d} deflate blocks per BGZF block: {best * 1e3:.2f} ms for 6000 blocks ({len(f) / 1e6:.0f} MB in), ok={ok}")
