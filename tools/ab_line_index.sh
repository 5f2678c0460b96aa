timeout 900 python -m pytest tests/test_gpu_vcf_parse.py tests/test_gpu_sam_parse.py tests/test_gpu_fastq_parse.py tests/test_gpu_fuzz_decode.py tests/test_typed_info.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
[ -f /tmp/e2e.fastq.gz ] || { tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6; }
cat /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz > /dev/null
for pass in 1 2; do for v in 2 1; do for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.fastq.gz fastq" "/tmp/e2e.vcf vcf"; do echo "== pass $pass index passes $v $spec"; EXON_HIP_LINE_INDEX_PASSES=$v python tools/time_pipeline_file.py $spec 6 2>&1 | tail -1 | cut -c1-110; done; done; done
