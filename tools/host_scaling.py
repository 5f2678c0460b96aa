"""How the CPU restatement (oracle/exon_oracle.c) scales with threads on this host, and what the host is: core count,
cgroup CPU quota, NUMA nodes.  Explains `cpu_baseline.parallel_speedup` of bench.py.  usage: python tools/host_scaling.py [rows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import Oracle  # noqa: E402


def sh(cmd):
    return os.popen(cmd + " 2>/dev/null").read().strip()


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 128_000_000
    print("nproc", sh("nproc"), "| os.cpu_count", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)))
    print("cgroup cpu.max:", sh("cat /sys/fs/cgroup/cpu.max") or sh("cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us"))
    print(sh("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz'"))
    print("loadavg", sh("cat /proc/loadavg"))
    orc = Oracle()
    t0 = time.perf_counter()
    af, av, q, qv, fid = orc.gen_c4(4, 0, n)
    print(f"gen {n} rows: {time.perf_counter() - t0:.2f}s")
    one = None
    for T in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if T > (os.cpu_count() or 1) and T != 1:
            break
        m = n if T >= 8 else n // 8
        best = 1e9
        for _ in range(3):
            _, _, _, t = orc.c4_cmp_avg_by_group(af[:m], av[:(m + 7) // 8], q[:m], qv[:(m + 7) // 8], fid[:m], orc.c4_filters(), 0.01, ">", threads=T)
            best = min(best, t.seconds_exec)
        rate = m / best / 1e6
        one = one or rate
        print(f"threads {T:4d}: {rate:9.1f} Mrows/s  speedup {rate / one:6.1f}  ({best * 1e3:.1f} ms exec on {m} rows)")
    # a STREAM-style read sum over the same columns with numpy (one thread) for scale
    t0 = time.perf_counter()
    s = float(af.sum(dtype=np.float64))
    dt = time.perf_counter() - t0
    print(f"numpy one-thread read of af: {af.nbytes / dt / 1e9:.1f} GB/s (sum {s:.3e})")


if __name__ == "__main__":
    main()
