#!/bin/bash
# staging-ring geometry sweep on one box, warm pipelines: base build vs ring 8x4 / 16x3 / 16x4 / 32x3
out=$1; base=$2
mkdir -p $out
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6
cat /tmp/e2e.vcf /tmp/e2e.vcf.gz /tmp/e2e.fastq /tmp/e2e.fastq.gz > /dev/null
for pass in 1 2; do
  for cfg in base 8x4 16x3 16x4 32x3; do
    for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.fastq fastq" "/tmp/e2e.fastq.gz fastq" "/tmp/e2e.vcf vcf"; do
      if [ $cfg = base ]; then
        r=$(EXON_HIP_LIB=$base python tools/time_pipeline_file.py $spec 6 2>&1 | tail -1)
      else
        r=$(EXON_HIP_RING_PIECE_MB=${cfg%x*} EXON_HIP_RING_PIECES=${cfg#*x} python tools/time_pipeline_file.py $spec 6 2>&1 | tail -1)
      fi
      echo "pass $pass $cfg $spec: $r" | tee -a $out/ring_sweep.log
    done
  done
done
