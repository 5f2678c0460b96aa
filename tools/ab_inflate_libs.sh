#!/bin/bash
# same-box A/B of one resident inflate launch per format between library builds: tools/ab_inflate_libs.sh <out> "<lib1> <lib2> ..." [passes]
# (a lib is a path for EXON_HIP_LIB, or "default")
out=$1; libs=$2; passes=${3:-2}
mkdir -p $out
export EXON_TIME_INFLATE_NO_HOST=1 EXON_HIP_INFLATE_PAR=0
for pass in $(seq $passes); do
  for spec in "vcf 28000000" "bam 10000000" "fastq 5000000"; do
    for lib in $libs; do
      echo "== pass $pass $spec $lib" >> $out/ab_libs.log
      if [ $lib = default ]; then timeout 600 python tools/time_inflate.py $spec 2>&1 | grep "crc=0" | tail -1 | cut -c1-100 >> $out/ab_libs.log
      else EXON_HIP_LIB=$lib timeout 600 python tools/time_inflate.py $spec 2>&1 | grep "crc=0" | tail -1 | cut -c1-100 >> $out/ab_libs.log; fi
    done
  done
done
cat $out/ab_libs.log
