#!/bin/bash
# round 3, GPU session 2: GROUP BY tiers -- parity, then 1e9-row bench lines per (G, dist), old global-only path as baseline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for spec in "64 uniform" "64 zipf" "4096 uniform" "4096 zipf" "100000 zipf" "100000 uniform"; do
  set -- $spec
  timeout 600 python bench.py --steps 10 --warmup 3 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/bench_g$1_$2.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2.json").read())
print("G=$1 $2", d["ms_per_step"], d["roofline"]["frac"], d["config"].get("groups"), d.get("parity","")[:40])
PY
done
for spec in "100000 zipf" "100000 uniform"; do
  set -- $spec
  EXON_HIP_K4_GLOBAL_ONLY=1 timeout 600 python bench.py --steps 5 --warmup 2 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/bench_g$1_$2_globalonly.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2_globalonly.json").read())
print("GLOBAL_ONLY G=$1 $2", d["ms_per_step"], d["roofline"]["frac"])
PY
done
tail -5 $O/bench.err
