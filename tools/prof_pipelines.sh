#!/bin/bash
# rocprofv3 --kernel-trace --stats of the three BGZF file -> answer pipelines (3 scans each of the 100 M-row .vcf.gz, the 20 M-read
# BAM and the 20 M-read .fastq.gz of tools/ab_pipes_env.sh), and their untraced times: tools/prof_pipelines.sh <out> [tag]
out=$1; tag=${2:-r5}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p $out
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
[ -f /tmp/e2e.fastq.gz ] || { tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6; }
[ -f /tmp/e2e.bam ] || { tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6; }
cat /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz /tmp/e2e.bam > /dev/null
for spec in "/tmp/e2e.vcf.gz vcf bgzf" "/tmp/e2e.bam bam bam" "/tmp/e2e.fastq.gz fastq fastq"; do
  set -- $spec
  python tools/time_pipeline_file.py $1 $2 6 >> $out/${tag}_pipes_untraced.log 2>&1
  rocprofv3 --kernel-trace --stats -d $out/tmp_$3 -o $3 --output-format csv -- python tools/time_pipeline_file.py $1 $2 3 > $out/${tag}_$3_traced.log 2>&1
  cp $out/tmp_$3/$3_kernel_stats.csv $out/${tag}_$3_pipeline_kernel_stats.csv
  rm -rf $out/tmp_$3
  echo "== $2"; head -12 $out/${tag}_$3_pipeline_kernel_stats.csv | cut -c1-150
done
cat $out/${tag}_pipes_untraced.log
