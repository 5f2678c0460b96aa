#!/bin/bash
# round 3, GPU session 12: K4 LDS tier with / without the uniform-key test on ONE box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
for u in 1 0 1 0; do
  for spec in "64 uniform" "4096 zipf"; do
    set -- $spec
    EXON_HIP_K4_UNIFORM=$u timeout 900 python bench.py --steps 10 --warmup 3 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/b.json
    python - <<PY
import json
d=json.loads(open("$O/b.json").read())
print("uniform_test=$u G=$1 $2", d["ms_per_step"], d["roofline"]["frac"])
PY
  done
done
timeout 600 python tools/time_skew.py 2e8 2>> $O/skew.err | grep '^{' | tee $O/skew.log
timeout 300 python tools/time_small.py c4:1e9 c3:1e9 >> $O/small.log 2>&1; cat $O/small.log
