out=gpurun_out/r5_wide9; mkdir -p $out
for spec in "vcf 28000000" "bam 10000000" "fastq 5000000"; do set -- $spec; bash tools/pmc_inflate.sh $out 3 $1 $2 > /dev/null 2>&1; done
for f in vcf bam fastq; do echo "== $f"; awk '{printf "%s %.4g | ", $1, $NF} END{print ""}' $out/pmc_${f}_flavor3.txt; done
