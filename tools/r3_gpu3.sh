#!/bin/bash
# round 3, GPU session 3: GROUP BY tiers after the instruction diet; J=2 vs J=4 tiles
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_typed_info.py tests/test_gpu_vcf_parse.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for j4 in 0 1; do
for spec in "64 uniform" "64 zipf" "4096 uniform" "100000 zipf"; do
  set -- $spec
  EXON_HIP_K4_OVF_J4=$j4 timeout 600 python bench.py --steps 10 --warmup 3 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/bench_g$1_$2_j4$j4.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2_j4$j4.json").read())
print("J4=$j4 G=$1 $2", d["ms_per_step"], d["roofline"]["frac"], d.get("parity","")[:30])
PY
done
done
timeout 300 python tools/time_small.py c4:125e6 c4:1e9 c2:1e7 c3:1e8 >> $O/small.log 2>&1; cat $O/small.log
