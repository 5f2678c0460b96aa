#!/usr/bin/env python3
"""Debug helper: inflate fuzz blocks one by one, report the ones that differ from zlib."""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import exon_amd
from test_gpu_inflate import bgzf_block, vcf_like
ctx = exon_amd.Context(0)
rng = np.random.default_rng(2026)
text = vcf_like(4000, seed=9)
bad = 0
for i in range(3000):
    kind = i % 6
    size = int(rng.integers(0, 65281)) if i % 11 else int(rng.integers(0, 40))
    if kind == 0:
        off = int(rng.integers(0, max(1, len(text) - size))); data = text[off:off + size]
    elif kind == 1:
        data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    elif kind == 2:
        data = rng.integers(0, 4, size, dtype=np.uint8).tobytes()
    elif kind == 3:
        data = (bytes(rng.integers(65, 70, 37, dtype=np.uint8)) * (size // 37 + 1))[:size]
    elif kind == 4:
        data = bytes([int(rng.integers(0, 256))]) * size
    else:
        a = rng.integers(0, 256, size, dtype=np.uint8); a[rng.random(size) < 0.9] = 65; data = a.tobytes()
    level = int(rng.choice([0, 1, 4, 6, 9]))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    if level == 0 and len(data) > 65000:
        data = data[:65000]
    blk = bgzf_block(data, level, strategy)
    try:
        got, _ = ctx.bgzf_inflate(blk)
        ok = got.tobytes() == data
        msg = "" if ok else "DIFF at %d" % next((j for j in range(min(len(data), len(got))) if got[j] != data[j]), -1)
    except Exception as e:
        ok, msg = False, str(e)[-60:]
    if not ok:
        bad += 1
        if bad <= 25:
            print(f"block {i}: kind {kind} size {size} level {level} strategy {strategy}: {msg}")
print("bad:", bad)
