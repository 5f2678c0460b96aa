#!/bin/bash
# developer sweep: every exon_amd/lib/var/lib_<lit>_<ring>.so on three payloads (best of the crc=0 runs)
make -C tools -s >/dev/null 2>&1
cp exon_amd/lib/libexon_hip.so /tmp/lib_keep.so
for f in exon_amd/lib/var/lib_*.so; do
  cp $f exon_amd/lib/libexon_hip.so
  for k in "vcf 9300000" "bam 3100000" "fastq 1500000"; do
    r=$(timeout 200 python tools/time_inflate.py $k 2>&1 | grep "crc=0" | tail -1)
    echo "$(basename $f) $r"
  done
done
cp /tmp/lib_keep.so exon_amd/lib/libexon_hip.so
