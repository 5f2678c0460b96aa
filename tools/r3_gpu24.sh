#!/bin/bash
# round 3, GPU session 24: phases of the plain-text VCF pipeline (PCIe-bound?) 
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s24; mkdir -p $O
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf
cat /tmp/e2e.vcf > /dev/null
EXON_HIP_PIPE_TRACE=1 python tools/trace_vcfgz.py /tmp/e2e.vcf 5 2>&1 | grep -v amdgpu.ids > $O/trace_plain.log
grep -E "^run|setup|teardown" $O/trace_plain.log
python tools/measure_h2d.py 2>&1 | grep -v amdgpu.ids | tail -8
