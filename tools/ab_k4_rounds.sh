#!/bin/bash
# same-box A/B of the headline kernel (config 4) between two trees: the repo (HEAD) and a copy of an older commit under _ab/<name>
# (git worktree add /tmp/wt <commit>; cp -r /tmp/wt _ab/<name>; make -C _ab/<name>/exon_amd/csrc).  tools/ab_k4_rounds.sh <out> <name> [passes]
out=$1; name=$2; passes=${3:-3}
mkdir -p $out
for rows in 1000000000 125000000; do
  for pass in $(seq $passes); do
    for tree in . _ab/$name; do
      echo "== rows $rows pass $pass tree $tree" >> $out/ab_k4.log
      (cd $tree && python bench.py --rows $rows --steps 30 --warmup 5 --no-cpu-baseline --no-extras $(grep -q -- "--no-pmc" bench.py && echo --no-pmc) 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])") >> $out/ab_k4.log 2>&1
    done
  done
done
cat $out/ab_k4.log
