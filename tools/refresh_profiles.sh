cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1/c4 -o c4 --output-format csv -- python bench.py --steps 20 --warmup 3 > gpurun_out/prof_r1/bench_c4.json 2>/dev/null
cp gpurun_out/prof_r1/c4/c4_kernel_stats.csv gpurun_out/prof_r1/c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1/k6 -o k6 --output-format csv -- python tools/time_k6.py 1e9 > gpurun_out/prof_r1/k6.log 2>/dev/null
cp gpurun_out/prof_r1/k6/k6_kernel_stats.csv gpurun_out/prof_r1/k6_kernel_stats.csv
tools/bin/gen_text bam 20000000 /tmp/e2e.ubam 100 && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6
cat > /tmp/bam_one.py <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
for rep in range(2):
    scan = exon_amd.Scan("/tmp/e2e.bam", "bam", gpu_parse=True)
    plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, 25, columns=(0, 1, 2))
    st = plan.open(); rows = st.consume(scan); st.finish(); st.close(); plan.close(); scan.close()
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1/bam -o bam --output-format csv -- python /tmp/bam_one.py > /dev/null 2>&1
cp gpurun_out/prof_r1/bam/bam_kernel_stats.csv gpurun_out/prof_r1/bam_pipeline_kernel_stats.csv
tail -1 gpurun_out/prof_r1/bench_c4.json | cut -c1-400
head -3 gpurun_out/prof_r1/k6_kernel_stats.csv | cut -c1-200
head -5 gpurun_out/prof_r1/bam_pipeline_kernel_stats.csv | cut -c1-160
