#!/bin/bash
# round 3, GPU session 40: literal pairs in the serial inflate kernel's table (EXON_HIP_INFLATE_PAIRS): correctness, then A/B on
# the file pipelines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q -k "pair" > $O/pytest_pairs.log 2>&1; echo "rc $?" >> $O/pytest_pairs.log; tail -5 $O/pytest_pairs.log
B=tools/bin
make -C tools >/dev/null 2>&1
$B/gen_text vcf 100000000 /tmp/e2e.vcf && $B/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6 && rm /tmp/e2e.vcf
$B/gen_text bam 20000000 /tmp/e2e.ubam && $B/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6 && rm /tmp/e2e.ubam
$B/gen_text fastq 20000000 /tmp/e2e.fastq && $B/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6 && rm /tmp/e2e.fastq
for rep in 1 2 3; do
for pairs in 0 1; do
  for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.bam bam" "/tmp/e2e.fastq.gz fastq"; do
    EXON_HIP_INFLATE_PAIRS=$pairs timeout 300 python tools/time_pipeline_file.py $spec 6 2>&1 | grep -v amdgpu.ids | sed "s/\$/ pairs=$pairs/" >> $O/ab.log
  done
done
done
cat $O/ab.log
