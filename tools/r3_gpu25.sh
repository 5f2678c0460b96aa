#!/bin/bash
# round 3, GPU session 25: CRC check on the consumer's stream (next inflate does not wait for it): end to end, tests, fuzz
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s25; mkdir -p $O
for spec in "vcf 100000000" "bam 20000000" "bcf 50000000" "fastq 20000000"; do
  timeout 600 python tools/time_bgzf_pipeline.py $spec 2>&1 | grep -v amdgpu.ids | grep -E "GPU inflate|all equal|GB text" >> $O/pipelines_e2e.log
done
cat $O/pipelines_e2e.log
EXON_HIP_PIPE_TRACE=1 python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 4 2>&1 | grep -E "^run|setup" > $O/trace.log; cat $O/trace.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "pipeline or scan or vcf or bam or bcf or fastq or sam or region or cram or inflate or bgzf or fuzz" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
FUZZ_SEED=41 timeout 900 python tools/fuzz_gpu_decode.py 80 > $O/fuzz.log 2>&1; tail -10 $O/fuzz.log
