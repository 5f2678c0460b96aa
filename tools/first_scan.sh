#!/bin/bash
# Cold (first scan of a fresh context in a fresh process) and warm phases of the 100 M-row .vcf.gz pipeline, plus the other
# formats' pipelines as a regression check of the staging ring.  Output: gpurun_out/<dir>/first_scan.log
out=${1:-gpurun_out/r4_first}
mkdir -p $out
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
cat /tmp/e2e.vcf.gz > /dev/null
for i in 1 2; do
  echo "== fresh process $i" >> $out/first_scan.log
  EXON_HIP_PIPE_TRACE=1 python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 4 >> $out/first_scan.log 2>&1
done
