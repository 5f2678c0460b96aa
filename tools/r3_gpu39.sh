#!/bin/bash
# round 3, GPU session 39: CRAM host decode with the records turned into columns on the decode threads (the serial thread only
# copies blocks); thread counts above 32
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s39; mkdir -p $O
nproc > $O/cram.log
for th in "" 16 32 48 64 96; do
  echo "EXON_HIP_CRAM_THREADS=${th:-default}" >> $O/cram.log
  CRAM_REUSE=1 EXON_HIP_CRAM_THREADS=$th CRAM_REPEAT=10 timeout 600 python tools/time_cram.py 1000000 2>&1 | grep -v amdgpu.ids | grep -E "decode|K3|records" >> $O/cram.log
done
cat $O/cram.log
