#!/usr/bin/env python3
"""End to end on BGZF files: file.vcf.gz / file.fastq.gz -> answer, GPU inflate + GPU parse vs host inflate + GPU parse vs all-host decode."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")
kind = sys.argv[1] if len(sys.argv) > 1 else "vcf"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
plain, gz = f"/tmp/e2e.{kind}", f"/tmp/e2e.{kind}.gz"
if kind == "bam":
    plain, gz = "/tmp/e2e.ubam", "/tmp/e2e.bam"
if kind == "bcf":
    plain, gz = "/tmp/e2e.ubcf", "/tmp/e2e.bcf"
subprocess.check_call([os.path.join(BIN, "gen_text"), kind, str(n), plain])
subprocess.check_call([os.path.join(BIN, "bgzip"), plain, gz, "6"])
open(gz, "rb").read(); open(plain, "rb").read()
tsize, csize = os.path.getsize(plain), os.path.getsize(gz)
ctx = exon_amd.Context(0)


def run(path, gpu_parse):
    if kind in ("vcf", "bcf"):
        scan = exon_amd.Scan(path, kind, info_field="AF", gpu_parse=gpu_parse)
        plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    elif kind == "bam":
        scan = exon_amd.Scan(path, "bam", gpu_parse=gpu_parse)
        plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, 25, columns=(0, 1, 2))
    else:
        scan = exon_amd.Scan(path, "fastq", gpu_parse=gpu_parse)
        plan = ctx.plan_qual_pos_hist(256, columns=(3,))
    st = plan.open()
    t = time.perf_counter()
    rows = st.consume(scan)
    counts, sums = st.finish()
    dt = time.perf_counter() - t
    st.close(); plan.close(); scan.close()
    return rows, np.sort(np.array(counts)), dt  # FILTER ids are interned in first-seen order, which differs by path


def report(label, path, gpu_parse, reps=3):
    best = None
    for _ in range(reps):
        rows, c, dt = run(path, gpu_parse)
        best = dt if best is None else min(best, dt)
    print(f"{label:34s}: {rows} rows in {best:.3f} s = {rows / best / 1e6:8.1f} Mrows/s, {tsize / best / 1e9:6.2f} GB/s of text ({csize / best / 1e9:.2f} GB/s of file)")
    return c


print(f"{kind}: {tsize / 1e9:.2f} GB text, {csize / 1e9:.2f} GB BGZF")
a = report("bgzf: GPU inflate + GPU parse", gz, True)
if kind in ("bam", "bcf"):
    c = report("bgzf: host inflate + host decode", gz, False, reps=2)
    print("all equal:", np.array_equal(a, c))
    sys.exit(0)
os.environ["EXON_HIP_GPU_INFLATE"] = "0"
b = report("bgzf: host inflate + GPU parse", gz, True, reps=2)
c = report("bgzf: host inflate + host decode", gz, False, reps=2)
del os.environ["EXON_HIP_GPU_INFLATE"]
d = report("plain text: GPU parse", plain, True)
print("all equal (VCF group ids may be permuted):", np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d))
