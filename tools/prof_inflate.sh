# rocprofv3 kernel stats of the BGZF pipeline (GPU inflate + CRC + VCF parse + K4) on a 50M-row synthetic .vcf.gz
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tools/bin/gen_text vcf 50000000 /tmp/p.vcf && tools/bin/bgzip /tmp/p.vcf /tmp/p.vcf.gz 6
cat > /tmp/pipe_one.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
for rep in range(2):
    scan = exon_amd.Scan("/tmp/p.vcf.gz", "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    st = plan.open()
    t = time.perf_counter()
    rows = st.consume(scan)
    counts, sums = st.finish()
    print(rows, time.perf_counter() - t)
    st.close(); plan.close(); scan.close()
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bgzf -o run --output-format csv -- python /tmp/pipe_one.py 2>&1 | grep -v "^[EW]2026" | tail -3
cat gpurun_out/prof_bgzf/run_kernel_stats.csv | cut -c1-200
