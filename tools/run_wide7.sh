export EXON_TIME_INFLATE_NO_HOST=1 EXON_HIP_INFLATE_PAR=0 EXON_HIP_INFLATE_FLAVOR=3
out=gpurun_out/r5_wide7; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu -k "not fresh_process" -p no:cacheprovider 2>&1 | tail -2 >> $out/log
for spec in "vcf 28000000" "bam 10000000" "fastq 5000000"; do
  for f in 1 3; do echo "== $spec flavor $f" >> $out/log; EXON_HIP_INFLATE_FLAVOR=$f timeout 600 python tools/time_inflate.py $spec 2>&1 | grep "crc=0" | tail -2 | cut -c1-100 >> $out/log; done
done
unset EXON_HIP_INFLATE_PAR
for mb in 80 64; do
  EXON_HIP_GPU_PARSE_SLAB_MB=$mb bash tools/ab_pipes_env.sh $out/slab$mb EXON_HIP_INFLATE_FLAVOR "3" 1 > /dev/null 2>&1
  echo "== slab $mb MB" >> $out/log; grep -E "best" $out/slab$mb/ab_pipes_env.log | cut -c1-120 >> $out/log
done
cat $out/log
