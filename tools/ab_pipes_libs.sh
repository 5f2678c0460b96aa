#!/bin/bash
# same-box A/B of two library builds on the warm file pipelines: tools/ab_pipes_libs.sh <out> <libA> <libB>
out=$1; a=$2; b=$3
mkdir -p $out
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6
tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6
cat /tmp/e2e.vcf /tmp/e2e.vcf.gz /tmp/e2e.fastq /tmp/e2e.fastq.gz /tmp/e2e.bam > /dev/null
for pass in 1 2; do
  for lib in $a $b; do
    for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.vcf vcf" "/tmp/e2e.bam bam" "/tmp/e2e.fastq.gz fastq" "/tmp/e2e.fastq fastq"; do
      echo "== pass $pass lib $(basename $lib) $spec" >> $out/ab_pipes.log
      EXON_HIP_LIB=$lib python tools/time_pipeline_file.py $spec 6 >> $out/ab_pipes.log 2>&1
    done
  done
done
grep -E "^==|best|consume" $out/ab_pipes.log
