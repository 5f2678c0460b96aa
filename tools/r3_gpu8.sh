#!/bin/bash
# round 3, GPU session 8: where the partitioned tier 3 spends its time (kernel trace), full -m gpu suite at HEAD
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s8; mkdir -p $O
for spec in "100000 uniform" "100000 zipf" "16000000 uniform"; do
  set -- $spec
  tag=g$1_$2
  rocprofv3 --kernel-trace --stats -d $O/tmp_$tag -o $tag --output-format csv -- python bench.py --groups $1 --group-dist $2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$tag.json 2> /dev/null
  cp $(find $O/tmp_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv; rm -rf $O/tmp_$tag
  python - <<PY
import csv
print("== $tag")
for r in csv.DictReader(open("$O/${tag}_kernel_stats.csv")):
    if "k4_" in r["Name"] or "fill" in r["Name"]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e6,4), "ms avg", round(float(r["MaxNs"])/1e6,4), "max")
PY
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
