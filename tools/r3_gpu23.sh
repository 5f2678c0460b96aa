#!/bin/bash
# round 3, GPU session 23: every BGZF pipeline end to end after the kernels-only inflate stream; refreshed pipeline kernel stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s23; mkdir -p $O
bash tools/refresh_profiles.sh r3 pipelines > $O/pipelines.log 2>&1; tail -5 $O/pipelines.log
cat gpurun_out/prof_r3/vcfgz_end_to_end.log gpurun_out/prof_r3/bam_end_to_end.log 2>/dev/null
mkdir -p $O/prof; cp gpurun_out/prof_r3/*pipeline_kernel_stats.csv gpurun_out/prof_r3/*end_to_end.log $O/prof/ 2>/dev/null
for spec in "vcf 100000000" "bam 20000000" "bcf 50000000" "fastq 20000000"; do
  timeout 600 python tools/time_bgzf_pipeline.py $spec 2>&1 | grep -v amdgpu.ids >> $O/pipelines_e2e.log
done
cat $O/pipelines_e2e.log
