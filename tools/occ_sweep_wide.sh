#!/bin/bash
# resident waves per CU against throughput for one symbol loop: tools/occ_sweep_wide.sh <out> <flavor> <kind> <rows>
# (EXON_HIP_INFLATE_PAD_LDS pads a workgroup's LDS; a CU's 160 KiB come in 1280-byte steps)
out=$1; f=$2; kind=$3; rows=$4
mkdir -p $out
export EXON_TIME_INFLATE_NO_HOST=1 EXON_HIP_INFLATE_PAR=0 EXON_HIP_INFLATE_FLAVOR=$f
for pad in 0 1536 3328 5376 8704 15616; do
  echo "== $kind flavor $f pad $pad" >> $out/occ.log
  EXON_HIP_INFLATE_PAD_LDS=$pad timeout 300 python tools/time_inflate.py $kind $rows 2>&1 | grep "crc=0" | tail -1 >> $out/occ.log
done
