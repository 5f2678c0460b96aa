#!/usr/bin/env python3
"""Step time of the configs at the sizes BASELINE.json states them (c2 @ 1e7, c3 @ 1e8, the 125e6-row shard of config 4 at
8 GPUs, ...) -- the launch-bound end of the path.  Usage: tools/time_small.py [kind:rows ...]; env EXON_HIP_FUSE_FOLD /
EXON_HIP_SHAPE select the variants being compared (the library reads them once per process)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import exon_amd  # noqa: E402

specs = sys.argv[1:] or ["c2:1e7", "c3:1e8", "c4:125e6", "c4:1e9"]
ctx = exon_amd.Context(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
tag = {k: os.environ.get(k) for k in ("EXON_HIP_FUSE_FOLD", "EXON_HIP_SHAPE") if os.environ.get(k) is not None}
for sp in specs:
    kind, rows = sp.split(":")
    rows = int(float(rows))
    best = None
    for rep in range(3):
        r = bench.time_config(ctx, kind, rows, steps=200 if rows <= 2e8 else 20, warmup=10)
        if best is None or r["ms_per_step"] < best["ms_per_step"]:
            best = r
    print(json.dumps({"kind": kind, **tag, **best}), flush=True)
