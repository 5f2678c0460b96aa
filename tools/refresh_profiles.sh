#!/bin/bash
# Regenerates every rocprofv3 summary the docs quote, at HEAD, on the GPU box:  tools/refresh_profiles.sh r2
#   (note: `bench.py` under --kernel-trace pays tens of microseconds per launch: its JSON is not a throughput figure for the per-batch c5 path's many
#   launches per step; the quoted bench lines come from unprofiled runs, profiles/<round>_bench_<wl>_1gpu.json)
#   gpurun_out/prof_<round>/<wl>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --workload <wl>`
#   gpurun_out/pmc_fetch|pmc_write/<wl>_counter_collection.csv   separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel trace only)
#   gpurun_out/prof_<round>/{bgzf,bam}_pipeline_kernel_stats.csv  file -> answer pipelines
# then tools/collect_profiles.py copies the summaries into profiles/<round>_* and rewrites profiles/traffic.json.
R=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=gpurun_out/prof_$R
mkdir -p $P gpurun_out/pmc_fetch gpurun_out/pmc_write
run_wl() {  # workload rows steps
  rocprofv3 --kernel-trace --stats -d $P/tmp_$1 -o $1 --output-format csv -- python bench.py --workload $1 --rows $2 --steps $3 --warmup 3 --no-extras --no-pmc > $P/bench_$1.json 2> $P/bench_$1.err
  cp $P/tmp_$1/$1_kernel_stats.csv $P/$1_kernel_stats.csv; rm -rf $P/tmp_$1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmc_$( [ $ctr = FETCH_SIZE ] && echo fetch || echo write )
    rm -rf $d/tmp_$1
    rocprofv3 --pmc $ctr --kernel-trace -d $d/tmp_$1 -o $1 --output-format csv -- python bench.py --workload $1 --rows $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > /dev/null 2>&1
    f=$(find $d/tmp_$1 -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $d/$1_counter_collection.csv
    rm -rf $d/tmp_$1
  done
  python tools/collect_profiles.py $R $1 $2 | tail -1
  tail -1 $P/bench_$1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
}
WHAT=${2:-"c4 c2 c3 c6 c5 pipelines"}   # second argument: a subset, e.g. "c5"
for w in $WHAT; do
  case $w in
    c5) run_wl c5 1e9 6 ;;
    c2|c3|c4|c6) run_wl $w 1e9 20 ;;
  esac
done
case " $WHAT " in *" pipelines "*) ;; *) exit 0 ;; esac
# file -> answer pipelines
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat > /tmp/vcfgz_one.py <<PY
import sys, os, time
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
for rep in range(3):
    t0 = time.perf_counter()
    scan = exon_amd.Scan("/tmp/e2e.vcf.gz", "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    st = plan.open(); rows = st.consume(scan); st.finish(); st.close(); plan.close(); scan.close()
    print("vcf.gz end to end", rows, "rows", round(time.perf_counter() - t0, 4), "s", flush=True)
PY
python /tmp/vcfgz_one.py > $P/vcfgz_end_to_end.log 2>&1
rocprofv3 --kernel-trace --stats -d $P/tmp_bgzf -o bgzf --output-format csv -- python /tmp/vcfgz_one.py > /dev/null 2>&1
cp $P/tmp_bgzf/bgzf_kernel_stats.csv $P/bgzf_pipeline_kernel_stats.csv; rm -rf $P/tmp_bgzf
tools/bin/gen_text bam 20000000 /tmp/e2e.ubam 100 && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6
cat > /tmp/bam_one.py <<PY
import sys, os, time
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
for rep in range(3):
    t0 = time.perf_counter()
    scan = exon_amd.Scan("/tmp/e2e.bam", "bam", gpu_parse=True)
    plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, 25, columns=(0, 1, 2))
    st = plan.open(); rows = st.consume(scan); st.finish(); st.close(); plan.close(); scan.close()
    print("bam end to end", rows, "rows", round(time.perf_counter() - t0, 4), "s", flush=True)
PY
python /tmp/bam_one.py > $P/bam_end_to_end.log 2>&1
rocprofv3 --kernel-trace --stats -d $P/tmp_bam -o bam --output-format csv -- python /tmp/bam_one.py > /dev/null 2>&1
cp $P/tmp_bam/bam_kernel_stats.csv $P/bam_pipeline_kernel_stats.csv; rm -rf $P/tmp_bam
tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6
cat /tmp/e2e.fastq.gz > /dev/null
python tools/time_pipeline_file.py /tmp/e2e.fastq.gz fastq 4 > $P/fastqgz_end_to_end.log 2>&1
rocprofv3 --kernel-trace --stats -d $P/tmp_fq -o fq --output-format csv -- python tools/time_pipeline_file.py /tmp/e2e.fastq.gz fastq 4 > /dev/null 2>&1
cp $P/tmp_fq/fq_kernel_stats.csv $P/fastq_pipeline_kernel_stats.csv; rm -rf $P/tmp_fq
cat $P/vcfgz_end_to_end.log $P/bam_end_to_end.log $P/fastqgz_end_to_end.log
