#!/bin/bash
# the reference's benchmark query shape (exon-benchmarks/src/main.rs:143-157: chrom, pos, id of the rows a region keeps) as batches
# from the GPU pipeline over a sorted synthetic .vcf.gz: views of the kept run vs the row-by-row gather.  usage: <outdir> [rows]
out=${1:-gpurun_out/region}; n=${2:-100000000}; mkdir -p $out; d=$(mktemp -d /tmp/rg.XXXX)
tools/bin/gen_text vcf $n $d/s.vcf && tools/bin/bgzip $d/s.vcf $d/s.vcf.gz 6 && rm $d/s.vcf
{
for rg in 1 7:50000000-100000000; do
  echo "## region $rg, + id (projection 1): the kept run as views"; EXON_TIME_REGION=$rg tools/bin/time_scan_next $d/s.vcf.gz vcf 4 1 2>&1 | grep -E "pass [13]|best"
  echo "## region $rg: row-by-row gather (EXON_HIP_EXPORT_GATHER=1)"; EXON_HIP_EXPORT_GATHER=1 EXON_TIME_REGION=$rg tools/bin/time_scan_next $d/s.vcf.gz vcf 3 1 2>&1 | grep -E "pass 1|best"
done
} | tee $out/region_batches.log
rm -rf $d
