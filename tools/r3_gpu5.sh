#!/bin/bash
# round 3, GPU session 5: chain-check / in-kernel zeroing parity; merge budget; pipeline profile
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bam_parse.py tests/test_gpu_bcf_parse.py tests/test_gpu_sam_parse.py tests/test_gpu_region_pushdown.py tests/test_gpu_fuzz_decode.py tests/test_typed_info.py tests/test_cram.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for spec in "c4 125e6" "c3 12.5e6" "c5 125e6"; do
  timeout 300 python tools/time_merge.py $spec 2>> $O/merge.err | grep '^{' >> $O/merge.log
done
cat $O/merge.log; tail -3 $O/merge.err
rocprofv3 --kernel-trace --stats -d $O/tmp_merge -o merge --output-format csv -- python tools/time_merge.py c4 125e6 2> /dev/null | grep '^{' > $O/merge_traced.json
cp $(find $O/tmp_merge -name "*kernel_stats.csv" | head -1) $O/merge_c4_kernel_stats.csv
rm -rf $O/tmp_merge
cut -c1-160 $O/merge_c4_kernel_stats.csv | head -12
bash tools/refresh_profiles.sh r3 pipelines > $O/pipelines.log 2>&1; tail -5 $O/pipelines.log
ls gpurun_out/prof_r3/ ; cat gpurun_out/prof_r3/vcfgz_end_to_end.log gpurun_out/prof_r3/bam_end_to_end.log 2>/dev/null
cut -c1-150 gpurun_out/prof_r3/bgzf_pipeline_kernel_stats.csv | head -12
cut -c1-150 gpurun_out/prof_r3/bam_pipeline_kernel_stats.csv | head -12
echo "--- EXON_HIP_STREAM_PRIORITY=0 (plain streams)"
EXON_HIP_STREAM_PRIORITY=0 python /tmp/vcfgz_one.py 2>&1 | tail -3
EXON_HIP_STREAM_PRIORITY=0 python /tmp/bam_one.py 2>&1 | tail -3
echo "--- priorities on, again"
python /tmp/vcfgz_one.py 2>&1 | tail -3
python /tmp/bam_one.py 2>&1 | tail -3
