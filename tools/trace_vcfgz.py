#!/usr/bin/env python3
"""Phases of the .vcf.gz -> GROUP BY pipeline, first run vs steady state (EXON_HIP_PIPE_TRACE=1 prints them on stderr).
usage: trace_vcfgz.py FILE [runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd  # noqa: E402

path = sys.argv[1]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t00 = time.perf_counter()
ctx = exon_amd.Context(0)
print("context %.1f ms" % ((time.perf_counter() - t00) * 1e3), flush=True)
for rep in range(runs):
    t0 = time.perf_counter()
    scan = exon_amd.Scan(path, "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    t1 = time.perf_counter()
    st = plan.open()
    rows = st.consume(scan)
    t2 = time.perf_counter()
    st.finish()
    st.close()
    plan.close()
    scan.close()
    t3 = time.perf_counter()
    print("run", rep, rows, "rows: open %.1f ms, consume %.1f ms, finish+close %.1f ms, total %.4f s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, t3 - t0), flush=True)
