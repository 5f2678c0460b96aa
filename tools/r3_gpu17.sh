#!/bin/bash
# round 3, GPU session 17: tier 3 without global cursors (per-workgroup histogram rows -> scan), prefetching scatter, unrolled aggregate
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s17; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_synth_goldens.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for sl in 4 2; do
for spec in "100000 uniform" "100000 zipf" "1000000 zipf" "16000000 uniform"; do
  set -- $spec
  EXON_HIP_K4_TAIL_SLICES_PER_CU=$sl timeout 900 python bench.py --steps 5 --warmup 2 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | grep '^{' | tail -1 > $O/bench_g$1_$2_sl$sl.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2_sl$sl.json").read())
print("slices/cu=$sl G=$1 $2", d["ms_per_step"], d["roofline"]["frac"], d.get("parity","")[:40])
PY
done
done
for spec in "100000 uniform" "16000000 uniform"; do
  set -- $spec
  tag=g$1_$2
  rocprofv3 --kernel-trace --stats -d $O/tmp_$tag -o $tag --output-format csv -- python bench.py --groups $1 --group-dist $2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  cp $(find $O/tmp_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv; rm -rf $O/tmp_$tag
  python - <<PY
import csv
print("== $tag")
for r in csv.DictReader(open("$O/${tag}_kernel_stats.csv")):
    if "k4_" in r["Name"] or "fill" in r["Name"]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e6,4), "ms avg")
PY
done
tail -3 $O/bench.err
