#!/bin/bash
# Warm end-to-end numbers of the four file pipelines (tools/time_bgzf_pipeline.py) -> <out>/pipelines_end_to_end.log
out=${1:-gpurun_out/r4_pipes}
mkdir -p $out
for spec in "vcf 100000000" "bam 20000000" "bcf 50000000" "fastq 20000000"; do
  python tools/time_bgzf_pipeline.py $spec >> $out/pipelines_end_to_end.log 2>&1
done
cat $out/pipelines_end_to_end.log
