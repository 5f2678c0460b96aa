# rocprofv3 kernel stats of the .vcf.gz file -> answer pipeline (two scans of a 100 M-row file) -> gpurun_out/prof_r1/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r1
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat > /tmp/vcfgz_one.py <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
for rep in range(2):
    scan = exon_amd.Scan("/tmp/e2e.vcf.gz", "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    st = plan.open(); rows = st.consume(scan); st.finish(); st.close(); plan.close(); scan.close()
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1/bgzf -o bgzf --output-format csv -- python /tmp/vcfgz_one.py > /dev/null 2>&1
cp gpurun_out/prof_r1/bgzf/bgzf_kernel_stats.csv gpurun_out/prof_r1/bgzf_pipeline_kernel_stats.csv
head -6 gpurun_out/prof_r1/bgzf_pipeline_kernel_stats.csv | cut -c1-170
