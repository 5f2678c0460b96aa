#!/usr/bin/env python3
"""The multi-GPU step on ONE GPU: config 4's 125e6-row shard (what each rank of an 8-GPU run holds) + exon_hip_merge_states
on a ONE-rank RCCL communicator (ncclAllGather of the 120-byte state + the rank-ordered fold) -- everything a rank does in a
step except waiting for its peers.  Budget for >= 6x at 8 GPUs: step(1e9 rows, 1 GPU) / 6.  Also tries the whole step as ONE
captured HIP graph (torch.cuda.CUDAGraph around the C-ABI launches and the RCCL call).
    python tools/time_merge.py [workload=c4] [rows=125e6]          (rocprofv3 --kernel-trace --stats -- python tools/time_merge.py ...)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
import exon_amd  # noqa: E402
from exon_amd.distributed import NativeComm  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "c4"
rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 125_000_000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = exon_amd.Context(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
wl = bench.Workload(ctx, kind, rows, 0, rows * 8 if kind != "c5" else rows)
comm = NativeComm(ctx)
V = wl.state.numel()
gathered = torch.zeros(V, dtype=torch.int64, device="cuda")
merged = torch.zeros_like(wl.state)


def timed(fn, steps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, e0.elapsed_time(e1) / steps


def step():
    wl.run()
    comm.merge(wl.state, wl.n_i64, gathered, merged)


def merge_only():
    comm.merge(wl.state, wl.n_i64, gathered, merged)


steps = 200 if rows <= 2e8 else 20
out = {"workload": kind, "rows": rows, "state_bytes": V * 8, "rccl_ranks": comm.count()[0]}
out["kernel_only_ms"] = round(timed(wl.run, steps)[1], 4)
out["merge_only_ms"] = round(timed(merge_only, steps)[1], 4)
wall, gpu = timed(step, steps)
out["step_ms_wall"], out["step_ms_gpu"] = round(wall, 4), round(gpu, 4)
ref = merged.clone()
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        step()
    merged.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(merged, ref), "graph replay produced a different state"
    wall, gpu = timed(g.replay, steps)
    out["graph_step_ms_wall"], out["graph_step_ms_gpu"] = round(wall, 4), round(gpu, 4)
except Exception as e:  # noqa: BLE001
    out["graph_error"] = repr(e)[:300]
print(json.dumps(out), flush=True)
comm.close()
dist.destroy_process_group()
