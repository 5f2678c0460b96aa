#!/usr/bin/env python3
"""Times decode-only (native scan -> device-layout batches) and end-to-end CLI queries on a synthetic VCF file."""
import ctypes as C
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd  # noqa: E402

path, rows = sys.argv[1], int(float(sys.argv[2]))
rel = C.CFUNCTYPE(None, C.POINTER(exon_amd._lib.ArrowArray))
for threads in [int(x) for x in sys.argv[3].split(",")]:
    os.environ["EXON_HIP_DECODE_THREADS"] = str(threads)
    t = time.time()
    s = exon_amd.Scan(path, "vcf", info_field="AF", batch_size=1 << 20)
    n = 0
    while True:
        a = s.next_raw()
        if a is None:
            break
        n += a.length
        rel(a.release)(C.byref(a))
    dt = time.time() - t
    s.close()
    print(f"decode only, {threads:3d} threads: {n} rows in {dt:.2f}s = {n / dt / 1e6:.1f} Mrows/s ({os.path.getsize(path) / dt / 1e9:.2f} GB/s of text)")
    assert n == rows
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exon_amd", "bin", "exon-hip-cli")
os.environ.pop("EXON_HIP_DECODE_THREADS", None)
t = time.time()
out = subprocess.run([cli, "-q", "-c", "SET exon.vcf_parse_info = true;" f"CREATE EXTERNAL TABLE v STORED AS VCF LOCATION '{path}';"
                      'SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" > 0.01 GROUP BY filter'], capture_output=True, text=True)
dt = time.time() - t
print(out.stdout, out.stderr)
print(f"CLI end to end (file -> decode -> pinned staging -> HBM -> K4): {dt:.2f}s = {rows / dt / 1e6:.1f} Mrows/s")
