#!/bin/bash
# same-box A/B of two library builds on the configs at their stated sizes: tools/ab_small.sh <out> <libA> <libB> [kind:rows ...]
out=$1; a=$2; b=$3; shift 3
specs=${@:-"c2:1e7 c3:1e8 c6:1e7 c4:125e6 c4:1e7"}
mkdir -p $out
for pass in 1 2 3; do
  for lib in $a $b; do
    EXON_HIP_LIB=$lib python tools/time_small.py $specs 2>&1 | grep kind | sed "s|^|pass $pass $(basename $lib) |" >> $out/ab_small.log
  done
done
cat $out/ab_small.log
