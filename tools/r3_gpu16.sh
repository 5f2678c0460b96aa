#!/bin/bash
# round 3, GPU session 16: first run of a file pipeline with the second buffer set allocated by a helper thread
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s16; mkdir -p $O
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat /tmp/e2e.vcf.gz > /dev/null
for rep in 1 2 3; do
  EXON_HIP_PIPE_TRACE=1 python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 3 2>&1 | grep -v amdgpu.ids >> $O/trace.log
done
grep -E "init:|^run|setup" $O/trace.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "pipeline or scan or vcf or bam or bcf or fastq or sam or region or cram" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
