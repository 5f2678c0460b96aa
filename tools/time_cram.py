#!/usr/bin/env python3
"""Host CRAM decoder throughput on a synthetic file from tests/cram_writer.py (gzip blocks, quality scores kept: CRAM_QUALITIES=0
leaves them out), and the file -> K3 pipeline.  EXON_HIP_CRAM_EAGER=1 expands every block as round 2 did."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cram_writer import synthetic_records, write_cram
import exon_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
refs = [("chrA", 30_000_000), ("chrB", 15_000_000), ("chrC", 4_000_000)]
path = "/tmp/time.cram"
reuse = os.environ.get("CRAM_REUSE") == "1" and os.path.exists(path)  # A/B runs over one file: skip the (slow) python writer
rep = int(os.environ.get("CRAM_REPEAT", "1"))
if reuse:
    n *= rep
    rep = 1
    os.environ["CRAM_REPEAT"] = "1"
t = time.time(); recs = [] if reuse else synthetic_records(n, refs, seed=5); reuse or write_cram(path, refs, recs, per_slice=5000, slices_per_container=2, seed=1, methods=(1,), qualities=os.environ.get("CRAM_QUALITIES", "1") != "0")
print(f"wrote {n} records, {os.path.getsize(path) / 1e6:.1f} MB in {time.time() - t:.1f} s (python writer)")
rep = int(os.environ.get("CRAM_REPEAT", "1"))
if rep > 1:  # the python writer is slow: repeat the data containers (each is self-contained) to get a file worth timing
    import struct
    raw = open(path, "rb").read()
    o = 26
    hdr_len = struct.unpack_from("<i", raw, o)[0]
    def hdr_size(o):  # container header: length i32, then ITF8 / LTF8 fields, landmarks, crc32
        q = o + 4
        def itf8(q):
            b = raw[q]
            return q + (1 if b < 0x80 else 2 if b < 0xC0 else 3 if b < 0xE0 else 4 if b < 0xF0 else 5)
        def ltf8(q):
            b = raw[q]; k = 0
            while k < 8 and (b << k) & 0x80: k += 1
            return q + 1 + k
        for _ in range(4): q = itf8(q)      # ref id, start, span, n records
        q = ltf8(q); q = ltf8(q)            # record counter, bases
        q = itf8(q)                         # n blocks
        nl_at = q; q = itf8(q)
        nl = raw[nl_at] if raw[nl_at] < 0x80 else None
        assert nl is not None
        for _ in range(nl): q = itf8(q)
        return q + 4 - o
    first = o + hdr_size(o) + hdr_len       # end of the file-header container
    eof = len(raw) - 38
    with open(path, "wb") as f:
        f.write(raw[:first]); [f.write(raw[first:eof]) for _ in range(rep)]; f.write(raw[eof:])
    n *= rep
    print(f"repeated the data containers {rep} x: {n} records, {os.path.getsize(path) / 1e6:.1f} MB")
for rep in range(3):
    t = time.time(); scan = exon_amd.Scan(path, "cram"); rows = sum(len(b) for b in scan); dt = time.time() - t; scan.close()
    print(f"host decode: {rows} records in {dt * 1e3:.1f} ms = {rows / dt / 1e6:.2f} M records/s")
ctx = exon_amd.Context(0)
for rep in range(3):
    t = time.time()
    scan = exon_amd.Scan(path, "cram")
    plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, 3, columns=(0, 1, 2))
    st = plan.open(); rows = st.consume(scan); counts, _ = st.finish(); st.close(); plan.close(); scan.close()
    dt = time.time() - t
    print(f"file -> K3: {rows} records in {dt * 1e3:.1f} ms = {rows / dt / 1e6:.2f} M records/s, counts {list(counts)}")
