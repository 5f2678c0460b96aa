#!/usr/bin/env python3
"""Random damage to the COMPRESSED bytes of BGZF members through exon_hip_bgzf_inflate (small launches: the lane-parallel
decoder; EXON_HIP_INFLATE_PAR=0: the serial one): the call must report the damaged block or return zlib's bytes -- never fault."""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import exon_amd
import test_gpu_inflate as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(13)
ctx = exon_amd.Context(0)
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
if kind == "text":
    text = T.vcf_like(60000)
else:  # "mixed": runs of every period, high-entropy bytes with long codes, text -- the wide loop's rare paths (overlap fold, long
    rng = np.random.default_rng(17)  # literal / distance codes decoded in the walk, symbols beyond a round's 64 bytes)
    parts = []
    for i in range(1500):
        d, total = int(rng.integers(1, 71)), int(rng.integers(3, 601))
        unit = rng.integers(0, 256, d, dtype=np.uint8).tobytes()
        parts.append(rng.integers(0, 256, int(rng.integers(0, 71)), dtype=np.uint8).tobytes() + (unit * (total // d + 1))[:total])
    skew = (rng.standard_normal(400_000) * 30 + 128).clip(0, 255).astype(np.uint8).tobytes()  # ~200 distinct bytes: codes of 4-14 bits
    text = b"".join(parts) + skew + T.vcf_like(8000)
good = T.bgzf_file(text)
want = np.frombuffer(text, np.uint8)
ok = rejected = 0
for it in range(n):
    b = bytearray(good)
    for _ in range(rnd.choice([1, 1, 2, 8, 32])):
        i = rnd.randrange(len(b))
        if rnd.random() < 0.5:
            b[i] = rnd.randrange(256)
        else:
            b[i] ^= 1 << rnd.randrange(8)
    try:
        blocks, nb, consumed, ob = exon_amd.bgzf_scan(bytes(b))
    except exon_amd.ExonHipError:
        rejected += 1
        continue
    try:
        got, _ = ctx.bgzf_inflate(bytes(b))
    except exon_amd.ExonHipError:
        rejected += 1
        continue
    assert len(got) == len(want) and np.array_equal(got, want), it  # CRC-32 passed: the bytes must be right
    ok += 1
print("inflate fuzz: identical", ok, "rejected", rejected, flush=True)
