#!/bin/bash
# native exon_hip_scan_next timings on synthetic files (tools/bin/time_scan_next): usage tools/time_scan_next_all.sh <outdir> [vcf_rows] [bam_reads]
out=${1:-gpurun_out/scan_next}; vr=${2:-100000000}; br=${3:-20000000}
mkdir -p $out; d=$(mktemp -d /tmp/sn.XXXX)
tools/bin/gen_text vcf $vr $d/s.vcf && tools/bin/bgzip $d/s.vcf $d/s.vcf.gz 6 && rm $d/s.vcf
tools/bin/gen_text bam $br $d/s.ubam 100 && tools/bin/bgzip $d/s.ubam $d/s.bam 6 && rm $d/s.ubam
{
echo "## $vr-row .vcf.gz: path columns"; tools/bin/time_scan_next $d/s.vcf.gz vcf 4 0 2>&1 | grep -E "pass|best"
echo "## + id, ref, alt (projection 7)"; tools/bin/time_scan_next $d/s.vcf.gz vcf 4 7 2>&1 | grep -E "pass|best"
echo "## $br-read BAM: path columns"; tools/bin/time_scan_next $d/s.bam bam 4 0 2>&1 | grep -E "pass|best"
echo "## + name, cigar, sequence (projection 7)"; tools/bin/time_scan_next $d/s.bam bam 4 7 2>&1 | grep -E "pass|best"
echo "## + name, cigar, sequence, quality_score (projection 15)"; tools/bin/time_scan_next $d/s.bam bam 3 15 2>&1 | grep -E "pass|best"
} | tee $out/scan_next.log
rm -rf $d
