#!/bin/bash
# round 3, GPU session 6: refresh every rocprofv3 summary the docs quote (kernel stats + FETCH/WRITE PMC of c2..c6 at 1e9
# rows), then the GROUP BY paths: kernel stats and instruction / atomic counters for 64 keys (LDS tier) and 1e5 keys (tail)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s6; mkdir -p $O
bash tools/refresh_profiles.sh r3 "c4 c2 c3 c6 c5" > $O/refresh.log 2>&1; tail -12 $O/refresh.log
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ATOMIC|SQ_INSTS_(VALU|LDS|SALU|VMEM)\b" | head -20 > $O/counters.txt; cat $O/counters.txt
for spec in "64 uniform" "100000 zipf" "100000 uniform"; do
  set -- $spec
  tag=g$1_$2
  rocprofv3 --kernel-trace --stats -d $O/tmp_$tag -o $tag --output-format csv -- python bench.py --groups $1 --group-dist $2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$tag.json 2> /dev/null
  cp $(find $O/tmp_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv; rm -rf $O/tmp_$tag
  for ctr in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum"; do
    name=$(echo $ctr | tr ' ' '+')
    rocprofv3 --pmc $ctr --kernel-trace -d $O/tmp_pmc -o p --output-format csv -- python bench.py --groups $1 --group-dist $2 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    f=$(find $O/tmp_pmc -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|k4_cmp_avg" "$f" | head -60 > $O/${tag}_pmc_$name.csv
    rm -rf $O/tmp_pmc
  done
done
ls -la $O | head -40
python - <<'PY'
import csv, glob, os
O = "gpurun_out/r3_s6"
for f in sorted(glob.glob(O + "/*_pmc_*.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(os.path.basename(f), {k: max(v) for k, v in agg.items()})
PY
