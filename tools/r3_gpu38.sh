#!/bin/bash
# round 3, GPU session 38: the lane-parallel / hybrid inflate on full pipeline slabs, re-measured now that the pipeline is
# inflate-bound (round 2 measured "moves nothing" while two small copies were what bound the loop)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s38; mkdir -p $O
B=tools/bin
make -C tools >/dev/null 2>&1
$B/gen_text vcf 100000000 /tmp/e2e.vcf && $B/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6 && rm /tmp/e2e.vcf
$B/gen_text bam 20000000 /tmp/e2e.ubam && $B/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6 && rm /tmp/e2e.ubam
$B/gen_text fastq 20000000 /tmp/e2e.fastq && $B/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6 && rm /tmp/e2e.fastq
ls -l /tmp/e2e.* > $O/files.log
for rep in 1 2; do
for cfg in "auto" "1" "2 0.3" "2 0.4" "2 0.5" "2 0.6"; do
  set -- $cfg
  for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.bam bam" "/tmp/e2e.fastq.gz fastq"; do
    if [ "$1" = auto ]; then
      timeout 300 python tools/time_pipeline_file.py $spec 6 2>&1 | grep -v amdgpu.ids >> $O/ab.log
    else
      EXON_HIP_INFLATE_PAR=$1 EXON_HIP_INFLATE_PAR_SERIAL_SHARE=${2:-0.4} timeout 300 python tools/time_pipeline_file.py $spec 6 2>&1 | grep -v amdgpu.ids >> $O/ab.log
    fi
  done
done
done
cat $O/ab.log
