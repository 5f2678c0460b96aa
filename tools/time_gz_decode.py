#!/usr/bin/env python3
"""Time exon_hip_gzip_stream_decode alone: a one-member plain-gzip file resident in HBM, one call per slab, no file pipeline.
usage: time_gz_decode.py {fastq|vcf} ROWS [slab_MB] [reps]   (EXON_HIP_LIB selects a variant build; CRC verification off so that
timing variants with wrong output still run)"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("EXON_HIP_GZ_VERIFY_CRC", "0")
import numpy as np  # noqa: E402
import exon_amd  # noqa: E402
from exon_amd.engine import DeviceBuffer  # noqa: E402
from time_plain_gzip import write_plain_gzip  # noqa: E402

kind, rows = sys.argv[1], int(float(sys.argv[2]))
slab = (int(sys.argv[3]) if len(sys.argv) > 3 else 128) << 20
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
text = f"/dev/shm/tg_{kind}_{rows}.{kind}"
gz = text + ".gz"
if not os.path.exists(gz):
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "gen_text"), kind, str(rows), text] + (["100", "0"] if kind == "fastq" else []))
    write_plain_gzip(text, gz)
    text_bytes = os.path.getsize(text)
    os.remove(text)
raw = np.fromfile(gz, np.uint8)
ctx = exon_amd.Context(0)
lib = ctx.lib
d_comp = DeviceBuffer(ctx, np.uint8, raw.size + 8192)
d_comp.copy_from(np.concatenate([raw, np.zeros(4096, np.uint8)]))
out_cap = 12 * slab
d_out = DeviceBuffer(ctx, np.uint8, out_cap)
for rep in range(reps):
    h = C.c_void_p()
    ctx._check(lib.exon_hip_gzip_stream_create(ctx.h, slab, int(os.environ.get("GZ_SCRATCH_X", "0")) * slab, C.byref(h)))
    pos, total, calls, t0 = 0, 0, 0, time.perf_counter()
    while True:
        n = min(slab, raw.size - pos)
        consumed, produced, end = C.c_int64(), C.c_int64(), C.c_int32()
        ctx._check(lib.exon_hip_gzip_stream_decode(h, None, d_comp.ptr + pos, n, int(pos + n == raw.size), d_out.ptr, out_cap, C.byref(consumed), C.byref(produced), C.byref(end)))
        pos += consumed.value
        total += produced.value
        calls += 1
        if end.value:
            break
        assert consumed.value > 0
    dt = time.perf_counter() - t0
    st = exon_amd._lib.GzipStats()
    lib.exon_hip_gzip_stream_get_stats(h, C.byref(st))
    lib.exon_hip_gzip_stream_destroy(h)
    print(f"{kind} {rows} rows: {raw.size / 1e6:.1f} MB -> {total / 1e6:.1f} MB in {dt * 1e3:.1f} ms = {total / dt / 1e9:.2f} GB/s out ({calls} calls, {st.chunks} chunks, {st.repairs} repairs) "
          f"[{os.environ.get('EXON_HIP_LIB', 'product build')}]", flush=True)
