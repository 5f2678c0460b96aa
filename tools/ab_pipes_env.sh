#!/bin/bash
# same-box A/B of the warm file pipelines under an environment switch: tools/ab_pipes_env.sh <out> VAR "v1 v2 ..." [passes]
out=$1; var=$2; vals=$3; passes=${4:-2}
mkdir -p $out
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
[ -f /tmp/e2e.fastq.gz ] || { tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6; }
[ -f /tmp/e2e.bam ] || { tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6; }
cat /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz /tmp/e2e.bam > /dev/null
for pass in $(seq $passes); do
  for v in $vals; do
    for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.bam bam" "/tmp/e2e.fastq.gz fastq"; do
      echo "== pass $pass $var=$v $spec" >> $out/ab_pipes_env.log
      env $var=$v python tools/time_pipeline_file.py $spec 6 >> $out/ab_pipes_env.log 2>&1
    done
  done
done
grep -E "^==|best" $out/ab_pipes_env.log
