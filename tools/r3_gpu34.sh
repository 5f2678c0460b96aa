#!/bin/bash
# round 3, GPU session 34: K5 chunk order staggered over workgroups: A/B of 1 / 2 / 8 / 64 start groups on the c5 bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s34; mkdir -p $O
cp exon_amd/lib/libexon_hip.so /tmp/libexon_hip_orig.so
for pass in 1 2; do
for st in 1 2 8 64; do
  cp exon_amd/lib/libexon_hip_st$st.so exon_amd/lib/libexon_hip.so
  python bench.py --workload c5 --no-cpu-baseline --steps 12 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger $st', d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/stagger.log
done
done
cp /tmp/libexon_hip_orig.so exon_amd/lib/libexon_hip.so
