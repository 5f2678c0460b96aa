"""Sanitizer fuzz of the rANS Nx16 decoder (round 3).  Streams of the test-side encoder (tests/rans_nx16_writer.py) with 1-4
random bytes changed ANYWHERE (flags, sizes, tables, states, words), truncated, or extended, through the product decoder built
with ASAN + UBSAN; and CRAM 3.1 files damaged behind the block CRC (harness-only macro EXON_CRAM_FUZZ_SKIP_CRC).
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iexon_amd/csrc -Iinclude tests/rans_nx16_harness.cpp -o /tmp/nx16h_asan -lz -ldl -lpthread
  g++ ... -DEXON_CRAM_FUZZ_SKIP_CRC tools/fuzz_host_asan.cpp -o /tmp/fz_nocrc -lz -lpthread -ldl
  FUZZ_SEED=1 python tools/fuzz_rans_nx16.py /tmp/nx16h_asan /tmp/fz_nocrc
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import rans_nx16_writer as nx  # noqa: E402
from cram_writer import synthetic_records, write_cram  # noqa: E402

seed = int(os.environ.get("FUZZ_SEED", "1"))
rng = np.random.default_rng(seed)
harness, fz = sys.argv[1], sys.argv[2]
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")

base = []
for data in (bytes(rng.integers(0, 256, 600, dtype=np.uint8)), bytes(rng.integers(0, 4, 2000, dtype=np.uint8)),
             bytes(np.repeat(rng.integers(0, 9, 200, dtype=np.uint8), rng.integers(1, 20, 200))),
             bytes(rng.choice([0, 1, 2, 60, 255], 1500, p=[.5, .2, .2, .05, .05]).astype(np.uint8))):
    for flags in (0x00, 0x01, 0x04, 0x05, 0x40, 0x41, 0x80, 0xC1, 0xC5, 0x08, 0x09, 0x20):
        for opt in (dict(), dict(o1_bits=10, code_table=True, code_rle_meta=False)):
            base.append((len(data), nx.encode(data, flags, **opt)))
cases = []
for _ in range(int(os.environ.get("FUZZ_STREAMS", "20000"))):
    n, s = base[int(rng.integers(0, len(base)))]
    b = bytearray(s)
    kind = int(rng.integers(0, 10))
    if kind == 0:
        b = b[:int(rng.integers(0, len(b)))]
    elif kind == 1:
        b += bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
    else:
        head = kind < 6  # half of the byte changes go to the first 48 bytes (flags, sizes, tables)
        for _k in range(int(rng.integers(1, 5))):
            at = int(rng.integers(0, min(len(b), 48) if head else len(b)))
            b[at] = int(rng.integers(0, 256))
    cases.append((n if rng.integers(0, 8) else int(rng.integers(0, 1 << 20)), bytes(b)))
inp = b"".join(struct.pack("<II", n, len(s)) + s for n, s in cases)
r = subprocess.run([harness], input=inp, capture_output=True, env=env)
if r.returncode != 0:
    print(r.stderr[-3000:].decode(errors="replace"))
ok = err = o = 0
for _ in cases:
    st, n = struct.unpack_from("<iI", r.stdout, o)
    o += 8 + n
    ok += st == 0
    err += st != 0
print(f"streams: {len(cases)} damaged, decoded {ok}, refused {err}, harness rc {r.returncode}", r.stderr[-400:].decode(errors="replace"))
assert r.returncode == 0 and o == len(r.stdout)

refs = [("chrA", 3_000_000), ("chrB", 1_500_000)]
with tempfile.TemporaryDirectory() as d:
    good = os.path.join(d, "good.cram")
    write_cram(good, refs, synthetic_records(3000, refs, seed=seed), per_slice=400, seed=seed, methods=(5,), qualities=True)
    raw = open(good, "rb").read()
    paths = []
    for i in range(int(os.environ.get("FUZZ_FILES", "600"))):
        b = bytearray(raw)
        for _k in range(int(rng.integers(1, 4))):
            b[int(rng.integers(26, len(b)))] = int(rng.integers(0, 256))
        p = os.path.join(d, f"f{i}.cram")
        open(p, "wb").write(bytes(b))
        paths.append(p)
    r = subprocess.run([fz, "cram"] + paths, capture_output=True, env=env)
    print("files:", r.stdout.decode().strip(), "rc", r.returncode, r.stderr[-300:].decode(errors="replace"))
    assert r.returncode == 0
