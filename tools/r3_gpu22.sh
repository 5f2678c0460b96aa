#!/bin/bash
# round 3, GPU session 22: does a longer inflate launch amortise its tail?  slab 64 / 128 / 256 MB with the kernels-only inflate stream
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s22; mkdir -p $O
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat /tmp/e2e.vcf.gz > /dev/null
for mb in 64 128 256 64; do
  echo "== slab $mb MB" >> $O/trace.log
  EXON_HIP_GPU_PARSE_SLAB_MB=$mb EXON_HIP_PIPE_TRACE=1 python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 4 2>&1 | grep -E "^run|setup" >> $O/trace.log
done
cat $O/trace.log
