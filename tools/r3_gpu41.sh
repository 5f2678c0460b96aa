#!/bin/bash
# round 3, GPU session 41: CRAM reader whose decode threads build the finished batches (pooled blocks): native drain by containers
# in flight, then file -> K3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s41; mkdir -p $O
g++ -std=c++17 -O2 -Iexon_amd/csrc -Iinclude tools/time_cram_native.cpp -o /tmp/time_cram_native -lz -lpthread -ldl
CRAM_REUSE=1 CRAM_REPEAT=10 timeout 600 python tools/time_cram.py 1000000 2>&1 | grep -v amdgpu.ids | grep -E "decode|K3|records" > $O/cram.log
for th in 8 16 32 48 64 96 128; do /tmp/time_cram_native /tmp/time.cram $th 8192 3 | tail -2 >> $O/cram.log; done
for th in 32 64 96; do
  echo "EXON_HIP_CRAM_THREADS=$th" >> $O/cram.log
  CRAM_REUSE=1 EXON_HIP_CRAM_THREADS=$th CRAM_REPEAT=10 timeout 600 python tools/time_cram.py 1000000 2>&1 | grep -v amdgpu.ids | grep -E "decode|K3" >> $O/cram.log
done
cat $O/cram.log
