#!/usr/bin/env python3
"""Throughput of the GPU BGZF inflate on a resident compressed slab (VCF text and FASTQ), next to zlib on the host."""
import os, subprocess, sys, time, zlib, gzip
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")
kind = sys.argv[1] if len(sys.argv) > 1 else "vcf"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000
level = sys.argv[3] if len(sys.argv) > 3 else "6"
plain, comp = f"/tmp/inf_bench.{kind}", f"/tmp/inf_bench.{kind}.gz"
plain, comp = f"/tmp/inf_bench.{kind}.{n}", f"/tmp/inf_bench.{kind}.{n}.{level}.gz"
if not (os.path.exists(plain) and os.path.exists(comp)):  # (kept between calls of one gpurun command: PMC passes, A/B runs)
    subprocess.check_call([os.path.join(BIN, "gen_text"), kind, str(n), plain])
    subprocess.check_call([os.path.join(BIN, "bgzip"), plain, comp, level])
raw = open(comp, "rb").read()
want = np.fromfile(plain, np.uint8)
ctx = exon_amd.Context(0)
for verify in (False, True):
    for rep in range(3):
        got, dt = ctx.bgzf_inflate(raw, verify_crc=verify)
        print(f"{kind} level {level} crc={int(verify)}: {len(raw) / 1e6:.0f} MB -> {len(got) / 1e6:.0f} MB in {dt * 1e3:.2f} ms = "
              f"{len(got) / dt / 1e9:.1f} GB/s out, {len(raw) / dt / 1e9:.1f} GB/s in; equal {np.array_equal(got, want)}")
if os.environ.get("EXON_TIME_INFLATE_NO_HOST"):
    sys.exit(0)
m = min(len(raw), 64 << 20)
t = time.perf_counter()
blocks, nb, consumed, ob = exon_amd.bgzf_scan(raw[:m])
ts = time.perf_counter() - t
t = time.perf_counter()
d = zlib.decompressobj(31)
o = 0
buf = raw[:consumed]
while buf:
    o += len(d.decompress(buf))
    buf = d.unused_data
    d = zlib.decompressobj(31)
tz = time.perf_counter() - t
print(f"host: block scan {consumed / ts / 1e9:.1f} GB/s; zlib 1 thread {o / tz / 1e6:.0f} MB/s out")
try:  # outcome counters of the lane-parallel path (EXON_HIP_INFLATE_PAR=1): index 0 = blocks decoded in parallel
    import ctypes
    st = (ctypes.c_uint32 * 32)()
    ctx.lib.exon_hip_bgzf_inflate_par_stats.restype = ctypes.c_int
    ctx.lib.exon_hip_bgzf_inflate_par_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    import exon_amd._lib as _L
    s0 = ctypes.c_void_p()
    if ctx.lib.exon_hip_bgzf_inflate_par_stats(None, st) == 0:
        print("par outcomes [0 = parallel, n = fallback reason]:", list(st))
except Exception as e:  # noqa: BLE001
    print("no par stats:", e)
