#!/bin/bash
# A/B of the high-cardinality GROUP BY tier: direct partition (default) vs compact -> scatter -> aggregate
# (EXON_HIP_K4_TAIL_SCATTER=1).  usage: tools/groupby_ab.sh <outdir>
out=${1:-gpurun_out/groupby_ab}; mkdir -p $out
for cfg in "100000 uniform" "100000 zipf" "500000 uniform" "1000000 zipf" "20000 uniform"; do
  set -- $cfg
  for sc in 0 1; do
    echo "## groups $1 $2 scatter=$sc" >> $out/ab.log
    EXON_HIP_K4_TAIL_SCATTER=$sc timeout 240 python bench.py --workload c4 --groups $1 --group-dist $2 --steps 10 --warmup 2 2>&1 | tee -a $out/raw.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('ms_per_step', round(d['ms_per_step'], 3), 'frac', round(r['frac'], 4), 'value', round(d['value'], 1))" >> $out/ab.log
  done
done
cat $out/ab.log
