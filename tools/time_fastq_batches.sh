#!/bin/bash
# exon_hip_scan_next over a synthetic FASTQ file: GPU pipeline vs the host reader.  usage: tools/time_fastq_batches.sh <outdir> [reads]
out=${1:-gpurun_out/fq}; n=${2:-20000000}; mkdir -p $out; d=$(mktemp -d /tmp/fq.XXXX)
tools/bin/gen_text fastq $n $d/s.fastq 100 && tools/bin/bgzip $d/s.fastq $d/s.fastq.gz 6
{
echo "## $n reads of 100 bases, BGZF: batches from the GPU pipeline"; EXON_HIP_PIPE_TRACE=1 tools/bin/time_scan_next $d/s.fastq.gz fastq 4 2>&1 | grep -E "pass|best|export" | tail -6
echo "## the same file: the host reader (EXON_TIME_HOST=1)"; EXON_TIME_HOST=1 tools/bin/time_scan_next $d/s.fastq.gz fastq 3 2>&1 | grep -E "pass|best"
echo "## plain text: GPU pipeline"; tools/bin/time_scan_next $d/s.fastq fastq 3 2>&1 | grep -E "pass|best"
} | tee $out/fastq_batches.log
rm -rf $d
