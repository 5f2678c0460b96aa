#!/bin/bash
# round 3, GPU session 11: uniform-key tests with back-off; K3 global path with wave aggregation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_synth_goldens.py tests/test_gpu_stream.py tests/test_gpu_bam_parse.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python tools/time_skew.py 2e8 2>> $O/skew.err | grep '^{' | tee $O/skew.log
timeout 300 python tools/time_small.py c4:1e9 c3:1e9 c3:1e8 c2:1e7 c4:125e6 >> $O/small.log 2>&1; cat $O/small.log
for spec in "64 uniform" "4096 zipf" "100000 zipf"; do
  set -- $spec
  timeout 900 python bench.py --steps 5 --warmup 2 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/bench_g$1_$2.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2.json").read())
print("G=$1 $2", d["ms_per_step"], d["roofline"]["frac"], d.get("parity","")[:30])
PY
done
