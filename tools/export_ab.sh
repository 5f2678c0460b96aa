#!/bin/bash
# batches from the GPU pipeline: a slab's copy under the next slab's inflate + parse (default) vs emitted at once (EXON_HIP_EXPORT_SYNC=1)
out=${1:-gpurun_out/export_ab}; mkdir -p $out; d=$(mktemp -d /tmp/ab.XXXX)
tools/bin/gen_text vcf 100000000 $d/s.vcf && tools/bin/bgzip $d/s.vcf $d/s.vcf.gz 6 && rm $d/s.vcf
tools/bin/gen_text bam 20000000 $d/s.ubam 100 && tools/bin/bgzip $d/s.ubam $d/s.bam 6 && rm $d/s.ubam
tools/bin/gen_text fastq 20000000 $d/s.fastq 100 && tools/bin/bgzip $d/s.fastq $d/s.fastq.gz 6 && rm $d/s.fastq
{
for sync in 0 1; do
  echo "## EXON_HIP_EXPORT_SYNC=$sync"
  for cfg in "s.vcf.gz vcf 0" "s.vcf.gz vcf 7" "s.bam bam 0" "s.bam bam 7" "s.bam bam 15" "s.fastq.gz fastq 0"; do
    set -- $cfg
    echo -n "$2 projection $3: "; EXON_HIP_EXPORT_SYNC=$sync tools/bin/time_scan_next $d/$1 $2 4 $3 2>&1 | grep best
  done
done
} | tee $out/export_ab.log
rm -rf $d
