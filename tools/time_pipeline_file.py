#!/usr/bin/env python3
"""Steady-state time of FILE -> answer through the GPU decode pipeline, on a file that already exists (for A/B runs of one file
under different environment switches).  usage: time_pipeline_file.py FILE {vcf|bcf|bam|fastq} [runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd  # noqa: E402

path, kind = sys.argv[1], sys.argv[2]
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = exon_amd.Context(0)
times = []
for rep in range(runs):
    t0 = time.perf_counter()
    if kind in ("vcf", "bcf"):
        scan = exon_amd.Scan(path, kind, info_field="AF", gpu_parse=True)
        plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    elif kind == "bam":
        scan = exon_amd.Scan(path, "bam", gpu_parse=True)
        plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, 25, columns=(0, 1, 2))
    else:
        scan = exon_amd.Scan(path, "fastq", gpu_parse=True)
        plan = ctx.plan_qual_pos_hist(256, columns=(3,))
    st = plan.open()
    t1 = time.perf_counter()
    rows = st.consume(scan)
    t2 = time.perf_counter()
    counts, sums = st.finish()
    st.close()
    plan.close()
    scan.close()
    t3 = time.perf_counter()
    times.append((t2 - t1, t3 - t0))
    check = int(sum(int(c) for c in counts))
warm = times[1:] if len(times) > 1 else times
print(f"{kind} {rows} rows, sum(counts) {check}: consume best {min(t[0] for t in warm) * 1e3:.1f} ms, median {sorted(t[0] for t in warm)[len(warm) // 2] * 1e3:.1f} ms; "
      f"open..close best {min(t[1] for t in warm) * 1e3:.1f} ms  [PAR={os.environ.get('EXON_HIP_INFLATE_PAR', 'auto')} share={os.environ.get('EXON_HIP_INFLATE_PAR_SERIAL_SHARE', '-')}]", flush=True)
