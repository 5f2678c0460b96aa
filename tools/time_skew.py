#!/usr/bin/env python3
"""How the LDS-atomic kernels behave on the key distributions real files have, not just the synthetic mix: a coordinate-sorted
BAM gives K3 long runs of ONE reference (all 64 lanes of a wave add to the same LDS counter), a VCF whose dominant FILTER list
got a dictionary id >= 4 does the same to K4's LDS tier.  Prints step times for: the bench's mix, runs of 1 M equal keys
(sorted), one constant key."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import exon_amd  # noqa: E402

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
ctx = exon_amd.Context(0)
torch.cuda.set_stream(torch.cuda.Stream())


def timed(wl, steps=20):
    for _ in range(3):
        wl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        wl.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


out = {"rows": rows}
wl = bench.Workload(ctx, "c3", rows, 0, rows)
out["k3_bench_mix_ms"] = round(timed(wl), 4)
idx = torch.arange(rows, device="cuda")
wl.ref[:rows] = ((idx >> 20) % 25).to(torch.int32)  # coordinate-sorted: runs of 2^20 reads per reference
wl.rv.fill_(0xFF)
out["k3_sorted_runs_ms"] = round(timed(wl), 4)
wl.ref[:rows] = 3
out["k3_one_reference_ms"] = round(timed(wl), 4)
del wl
for g, tag in ((64, "k4_lds_tier"), (5, "k4_registers")):
    wl = bench.Workload(ctx, "c4", rows, 0, rows, groups=g if g != 5 else 5)
    out[tag + "_bench_mix_ms"] = round(timed(wl), 4)
    if g != 5:
        wl.fid[:rows] = 10  # one hot key that lives in the LDS tier
        out[tag + "_one_hot_key_ms"] = round(timed(wl), 4)
        wl.fid[:rows] = ((idx >> 20) % (g - 4) + 4).to(torch.int32)
        out[tag + "_sorted_runs_ms"] = round(timed(wl), 4)
    del wl
print(json.dumps(out), flush=True)
