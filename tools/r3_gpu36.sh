#!/bin/bash
# round 3, GPU session 36: same-box A/B of the last-round dealing (old = per-tile, new = per wave-tile), shapes forced and default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s36; mkdir -p $O
cp exon_amd/lib/libexon_hip.so /tmp/keep.so
for pass in 1 2; do
for v in old new; do
  cp exon_amd/lib/libexon_hip_$v.so exon_amd/lib/libexon_hip.so
  for sh in "" 1 2; do
    EXON_HIP_SHAPE=$sh python tools/time_small.py c2:1e7 c3:1e8 c4:125e6 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', 'shape=$sh', d['kind'], d['rows'], d['ms_per_step'], d['frac'])" | tee -a $O/ab.log
  done
done
done
cp /tmp/keep.so exon_amd/lib/libexon_hip.so
