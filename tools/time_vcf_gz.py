import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import exon_amd
n = 100_000_000
subprocess.check_call(["tools/bin/gen_text", "vcf", str(n), "/tmp/e2e.vcf"])
subprocess.check_call(["tools/bin/bgzip", "/tmp/e2e.vcf", "/tmp/e2e.vcf.gz", "6"])
open("/tmp/e2e.vcf.gz", "rb").read()
ctx = exon_amd.Context(0)
for rep in range(4):
    scan = exon_amd.Scan("/tmp/e2e.vcf.gz", "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    st = plan.open()
    t = time.perf_counter()
    rows = st.consume(scan)
    st.finish()
    dt = time.perf_counter() - t
    st.close(); plan.close(); scan.close()
    print(f"vcf.gz: {rows} rows in {dt:.3f} s", file=sys.stderr)
