#!/usr/bin/env python3
"""PCIe-inclusive rate of the config-4 path from a PYTHON producer: pyarrow batches -> exon_hip_stream_push (pinned staging,
async H2D, double-buffered) -> fused kernel, with a parity check against the oracle.  pyarrow export + ctypes cost ~10 us per
push, so small batches measure Python here; tools/measure_h2d_native.cpp is the producer the quoted numbers come from.
Never the bench `value` (that one starts with data in HBM)."""
import os
import sys
import time

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd  # noqa: E402
from oracle import Oracle  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 64_000_000
batch = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4 << 20
orc = Oracle()
af, av, q, qv, fid = orc.gen_c4(4, 0, n)
avb = np.unpackbits(av, bitorder="little")[:n].astype(bool)
qvb = np.unpackbits(qv, bitorder="little")[:n].astype(bool)
rb = pa.record_batch({"af": pa.array(af, mask=~avb), "qual": pa.array(q, mask=~qvb),
                      "filter": pa.DictionaryArray.from_arrays(pa.array(fid), pa.array(orc.c4_filters()))})
batches = [rb.slice(i, batch) for i in range(0, n, batch)]
ctx = exon_amd.Context(0)
plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5)
st = plan.open()  # one stream, reused: the staging slots are allocated once per partition
for rep in range(3):
    st.reset()
    t0 = time.perf_counter()
    for b in batches:
        st.push(b)
    counts, sums = st.finish()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n / dt / 1e6:.1f} Mrows/s  ({n * 12.25 / dt / 1e9:.2f} GB/s of device-layout bytes, batch {batch} rows)")
st.close()
s, cn, cr, _ = orc.c4_cmp_avg_by_group(af, av, q, qv, fid, orc.c4_filters(), 0.01, ">")
assert np.array_equal(counts[5:], cr) and np.array_equal(counts[:5], cn) and np.allclose(sums, s, rtol=1e-6)
print("parity ok")
