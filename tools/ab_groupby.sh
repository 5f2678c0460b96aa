#!/bin/bash
# GROUP BY beyond the LDS table, same box: tier 3 grouped inside the main kernel (1) vs compact -> scatter -> aggregate (0)
# tools/ab_groupby.sh <out> ["G:dist ..."]
out=$1; specs=${2:-"100000:uniform 100000:zipf 500000:uniform 1000000:zipf 64:uniform"}
mkdir -p $out
for sp in $specs; do
  G=${sp%%:*}; D=${sp##*:}
  for b in 1 0; do
    echo "== groups $G $D binned=$b" >> $out/groupby.log
    EXON_HIP_K4_TAIL_BINNED=$b timeout 600 python bench.py --workload c4 --groups $G --group-dist $D --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d.get('parity','')[:40])" >> $out/groupby.log 2>&1
  done
done
cat $out/groupby.log
