#!/bin/bash
# round 3, GPU session 15: what the first run of a file pipeline pays for (allocation by allocation), slab sizes 64/32/16 MB
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s15; mkdir -p $O
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
ls -l /tmp/e2e.vcf /tmp/e2e.vcf.gz > $O/trace.log
cat /tmp/e2e.vcf.gz > /dev/null
for mb in 64 32 16; do
  echo "== slab $mb MB" >> $O/trace.log
  EXON_HIP_GPU_PARSE_SLAB_MB=$mb EXON_HIP_PIPE_TRACE=1 python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 4 2>&1 | grep -v amdgpu.ids >> $O/trace.log
done
cat $O/trace.log
