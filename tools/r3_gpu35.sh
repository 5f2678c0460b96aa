#!/bin/bash
# round 3, GPU session 35: last partial round of tiles dealt by wave-tile: parity of K2/K3/K4/K6, stated-size timings, shape A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s35; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "not k5 and not inflate and not fuzz and not cram" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python tools/time_small.py c2:1e7 c3:1e8 c4:125e6 c4:1e9 c3:1e9 c2:1e9 2>&1 | grep -v amdgpu | tee $O/small_default.log
for sh in 1 2; do EXON_HIP_SHAPE=$sh python tools/time_small.py c2:1e7 c3:1e8 c4:125e6 2>&1 | grep -v amdgpu | tee $O/small_shape$sh.log; done
