cd $GRAFT_REPO_ROOT
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat > /tmp/sw.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
open("/tmp/e2e.vcf.gz","rb").read()
for rep in range(3):
    scan = exon_amd.Scan("/tmp/e2e.vcf.gz", "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    st = plan.open()
    t = time.perf_counter()
    rows = st.consume(scan)
    counts, sums = st.finish()
    dt = time.perf_counter() - t
    st.close(); plan.close(); scan.close()
    print(os.environ.get("EXON_HIP_GPU_PARSE_SLAB_MB"), rows, f"{dt:.3f} s")
PY
for mb in 64; do EXON_HIP_PIPE_TRACE=1 EXON_HIP_GPU_PARSE_SLAB_MB=$mb python /tmp/sw.py; done
