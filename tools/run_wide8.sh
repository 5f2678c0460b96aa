out=gpurun_out/r5_wide8; mkdir -p $out
EXON_HIP_INFLATE_PAR=0 timeout 900 python tools/fuzz_gpu_inflate.py 600 > $out/fuzz_serial_wide.log 2>&1; tail -2 $out/fuzz_serial_wide.log
timeout 600 python tools/fuzz_gpu_inflate.py 300 > $out/fuzz_auto.log 2>&1; tail -2 $out/fuzz_auto.log
for spec in "vcf 28000000" "bam 10000000" "fastq 5000000"; do set -- $spec; bash tools/pmc_inflate.sh $out 3 $1 $2 > /dev/null 2>&1; done
for f in vcf bam fastq; do echo "== $f"; awk '{printf "%s %.4g | ", $1, $NF} END{print ""}' $out/pmc_${f}_flavor3.txt; done
