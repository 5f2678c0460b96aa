#!/bin/bash
# round 3, GPU session 9: key distributions real files have (sorted runs, one hot key); tier 3 after the faster head fold
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tier or 4096 or global" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python tools/time_skew.py 2e8 2>> $O/skew.err | grep '^{' | tee $O/skew.log
for spec in "100000 zipf" "100000 uniform"; do
  set -- $spec
  timeout 900 python bench.py --steps 5 --warmup 2 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/bench_g$1_$2.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2.json").read())
print("G=$1 $2", d["ms_per_step"], d["roofline"]["frac"], d.get("parity","")[:30])
PY
done
tail -3 $O/skew.err
