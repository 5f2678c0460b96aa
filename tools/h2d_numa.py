#!/usr/bin/env python3
"""H2D bandwidth of a pinned buffer (run under numactl --membind=N to see the NUMA effect)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
h.fill_(1)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2):
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D {5 * n / (time.perf_counter() - t) / 1e9:.1f} GB/s")
