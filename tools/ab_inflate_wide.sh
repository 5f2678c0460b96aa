#!/bin/bash
# the wide symbol loop (EXON_HIP_INFLATE_FLAVOR=3) against the vector loop (1): byte-exactness on tests/test_gpu_inflate.py with the
# serial kernels forced, then one resident launch per format.  tools/ab_inflate_wide.sh <out> [vcf rows] [quick]
out=$1; rows=${2:-14000000}
mkdir -p $out
if [ -z "$3" ]; then
  echo "== tests, flavor 3" >> $out/wide.log
  EXON_HIP_INFLATE_PAR=0 EXON_HIP_INFLATE_FLAVOR=3 timeout 900 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu -k "not fresh_process" -p no:cacheprovider 2>&1 | tail -15 >> $out/wide.log
fi
for spec in "vcf $rows" "bam 5000000" "fastq 2500000"; do
  for f in 1 3; do
    echo "== $spec flavor $f" >> $out/wide.log
    EXON_HIP_INFLATE_PAR=0 EXON_HIP_INFLATE_FLAVOR=$f timeout 300 python tools/time_inflate.py $spec 2>&1 | grep -E "crc=0" >> $out/wide.log
  done
done
cat $out/wide.log
