#!/bin/bash
# symbol loop on the scalar unit (0), the vector unit (1), both side by side (2): byte-exactness (tests/test_gpu_inflate.py with
# the serial kernel forced) and throughput of one resident launch per format.  tools/ab_inflate_flavor.sh <out> [rows]
out=$1; rows=${2:-5000000}
mkdir -p $out
for f in 1 2; do
  echo "== tests, flavor $f" >> $out/flavor.log
  EXON_HIP_INFLATE_PAR=0 EXON_HIP_INFLATE_FLAVOR=$f timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu -k "not fresh_process" -p no:cacheprovider 2>&1 | tail -3 >> $out/flavor.log
done
for kind in vcf bam fastq; do
  for f in 0 1 2; do
    echo "== $kind flavor $f" >> $out/flavor.log
    EXON_HIP_INFLATE_PAR=0 EXON_HIP_INFLATE_FLAVOR=$f timeout 300 python tools/time_inflate.py $kind $rows 2>&1 | grep -E "crc=|equal" >> $out/flavor.log
  done
done
cat $out/flavor.log
