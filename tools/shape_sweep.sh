#!/bin/bash
# step time of K2 / K3 / K6 / K4 under the three launch shapes at mid sizes (input for pick_shape): tools/shape_sweep.sh <out>
out=$1; mkdir -p $out
for sh in 1 2; do
  EXON_HIP_SHAPE=$sh python tools/time_small.py c2:1e7 c2:1.5e7 c2:2e7 c2:3e7 c2:5e7 c2:1e8 c3:2e7 c3:5e7 c3:1e8 c3:2e8 c6:1e7 c6:2e7 c6:5e7 c4:2e7 c4:5e7 c4:125e6 2>&1 | grep kind >> $out/shape_sweep.log
done
cat $out/shape_sweep.log
