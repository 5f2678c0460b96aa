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
t = time.time(); recs = synthetic_records(n, refs, seed=5); write_cram(path, refs, recs, per_slice=5000, slices_per_container=2, seed=1, methods=(1,), qualities=os.environ.get("CRAM_QUALITIES", "1") != "0")
print(f"wrote {n} records, {os.path.getsize(path) / 1e6:.1f} MB in {time.time() - t:.1f} s (python writer)")
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
