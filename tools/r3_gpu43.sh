#!/bin/bash
# round 3, GPU session 43: the CRAM reader on a file long enough for a steady state (50 M records, 3.4 GB)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s43; mkdir -p $O
g++ -std=c++17 -O2 -Iexon_amd/csrc -Iinclude tools/time_cram_native.cpp -o /tmp/time_cram_native -lz -lpthread -ldl
CRAM_REUSE=1 CRAM_REPEAT=50 timeout 900 python tools/time_cram.py 1000000 2>&1 | grep -v amdgpu.ids | grep -E "decode|K3|records" > $O/cram.log
for th in 16 32 64 96 128; do EXON_HIP_CRAM_TRACE=1 /tmp/time_cram_native /tmp/time.cram $th 8192 2 >> $O/cram.log 2>&1; done
for th in 32 96; do
  echo "EXON_HIP_CRAM_THREADS=$th" >> $O/cram.log
  CRAM_REUSE=1 EXON_HIP_CRAM_THREADS=$th CRAM_REPEAT=50 timeout 600 python tools/time_cram.py 1000000 2>&1 | grep -v amdgpu.ids | grep -E "K3" >> $O/cram.log
done
cat $O/cram.log
