#!/usr/bin/env python3
"""exon-hip-cli end to end (process start + HIP init included) on 100 M-row VCF text / .vcf.gz / .bcf and a 33 M-read BAM."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN, CLI = os.path.join(ROOT, "tools", "bin"), os.path.join(ROOT, "exon_amd", "bin", "exon-hip-cli")
n = sys.argv[1] if len(sys.argv) > 1 else "100000000"
subprocess.check_call([os.path.join(BIN, "gen_text"), "vcf", n, "/tmp/c.vcf"])
subprocess.check_call([os.path.join(BIN, "bgzip"), "/tmp/c.vcf", "/tmp/c.vcf.gz", "6"])
subprocess.check_call([os.path.join(BIN, "gen_text"), "bcf", n, "/tmp/c.ubcf"])
subprocess.check_call([os.path.join(BIN, "bgzip"), "/tmp/c.ubcf", "/tmp/c.bcf", "6"])
subprocess.check_call([os.path.join(BIN, "gen_text"), "bam", str(int(float(n)) // 3), "/tmp/c.ubam", "100"])
subprocess.check_call([os.path.join(BIN, "bgzip"), "/tmp/c.ubam", "/tmp/c.bam", "6"])
for f in ("/tmp/c.vcf", "/tmp/c.vcf.gz", "/tmp/c.bcf", "/tmp/c.bam"):
    open(f, "rb").read()
q4 = "SET exon.vcf_parse_info = true; CREATE EXTERNAL TABLE v STORED AS {fmt} LOCATION '{p}'{opt}; SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info.\"AF\" > 0.01 GROUP BY filter"
q3 = "CREATE EXTERNAL TABLE b STORED AS BAM LOCATION '/tmp/c.bam'; SELECT reference, COUNT(*) FROM b WHERE flag & 1284 = 0 AND CAST(mapping_quality AS INT) >= 30 GROUP BY reference"
cases = [("VCF text", q4.format(fmt="VCF", p="/tmp/c.vcf", opt="")), ("VCF .gz", q4.format(fmt="VCF", p="/tmp/c.vcf.gz", opt=" OPTIONS (compression gzip)")),
         ("BCF", q4.format(fmt="BCF", p="/tmp/c.bcf", opt="")), ("BAM config 3", q3)]
for name, sql in cases:
    for env_name, env in (("GPU decode", {}), ("host decode", {"EXON_HIP_GPU_PARSE": "0"})):
        best = None
        for _ in range(2):
            t = time.perf_counter()
            r = subprocess.run([CLI, "-c", sql], capture_output=True, text=True, env={**os.environ, **env})
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        tail = r.stdout.strip().splitlines()[-2][:60] if r.stdout.strip() else r.stderr[:80]
        print(f"{name:13s} {env_name:11s}: {best:.2f} s  rc={r.returncode}  {tail}")
