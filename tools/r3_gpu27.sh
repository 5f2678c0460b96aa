#!/bin/bash
# round 3, GPU session 27: K5 path A with a rolling load window (L) and without a branch per dword (M) against the shipped K
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s27; mkdir -p $O
for L in 100 148 64; do TUNE_K5_ONLY_IJ=1 timeout 300 tools/bin/tune_k5 1e8 $L 2>&1 | grep -v amdgpu.ids >> $O/tune_k5.log; done
cat $O/tune_k5.log
