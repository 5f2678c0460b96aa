#!/bin/bash
# same-box A/B of library builds x an environment switch on the warm file pipelines:
#   tools/ab_pipes_lib_env.sh <out> "<libA> <libB>" VAR "v1 v2"
out=$1; libs=$2; var=$3; vals=$4
mkdir -p $out
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
[ -f /tmp/e2e.fastq.gz ] || { tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6; }
[ -f /tmp/e2e.bam ] || { tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6; }
cat /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz /tmp/e2e.bam > /dev/null
for pass in 1 2; do
  for lib in $libs; do
    for v in $vals; do
      for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.bam bam" "/tmp/e2e.fastq.gz fastq"; do
        echo "== pass $pass $(basename $lib) $var=$v $spec" >> $out/ab.log
        env EXON_HIP_LIB=$lib $var=$v python tools/time_pipeline_file.py $spec 6 >> $out/ab.log 2>&1
      done
    done
  done
done
grep -E "^==|best" $out/ab.log | paste - - | cut -c1-175
