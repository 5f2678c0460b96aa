#!/bin/bash
# round 3, GPU session 42: where the CRAM reader's wall time goes on the 256-core host (EXON_HIP_CRAM_TRACE)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s42; mkdir -p $O
g++ -std=c++17 -O2 -Iexon_amd/csrc -Iinclude tools/time_cram_native.cpp -o /tmp/time_cram_native -lz -lpthread -ldl
CRAM_REUSE=1 CRAM_REPEAT=10 timeout 600 python tools/time_cram.py 1000000 2>&1 | grep -v amdgpu.ids | grep -E "decode|K3|records" > $O/cram.log
for th in 8 32 64 128; do EXON_HIP_CRAM_TRACE=1 /tmp/time_cram_native /tmp/time.cram $th 8192 3 >> $O/cram.log 2>&1; done
echo "--- taskset 64 cpus" >> $O/cram.log
EXON_HIP_CRAM_TRACE=1 taskset -c 0-63 /tmp/time_cram_native /tmp/time.cram 64 8192 3 >> $O/cram.log 2>&1
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core" >> $O/cram.log
cat $O/cram.log
