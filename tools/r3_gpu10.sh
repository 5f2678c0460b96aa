#!/bin/bash
# round 3, GPU session 10: uniform-key fast paths (sorted BAMs, hot LDS-tier key), BCF device lists, full suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s10; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python tools/time_skew.py 2e8 2>> $O/skew.err | grep '^{' | tee $O/skew.log
timeout 300 python tools/time_small.py c3:1e9 c3:1e8 c4:1e9 c2:1e7 >> $O/small.log 2>&1; cat $O/small.log
for spec in "64 uniform" "4096 zipf" "100000 zipf" "100000 uniform"; do
  set -- $spec
  timeout 900 python bench.py --steps 5 --warmup 2 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/bench_g$1_$2.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2.json").read())
print("G=$1 $2", d["ms_per_step"], d["roofline"]["frac"], d.get("parity","")[:30])
PY
done
