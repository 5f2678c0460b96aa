#!/bin/bash
# round 3, GPU session 7: partitioned tier 3 (compaction -> scatter by id range -> LDS table per range) vs global atomics
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s7; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_synth_goldens.py tests/test_gpu_stream.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
for mode in 0 1; do
for spec in "100000 zipf" "100000 uniform" "1000000 zipf" "16000000 uniform"; do
  set -- $spec
  EXON_HIP_K4_TAIL_ATOMICS=$mode timeout 900 python bench.py --steps 5 --warmup 2 --groups $1 --group-dist $2 --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/bench_g$1_$2_atomics$mode.json
  python - <<PY
import json
d=json.loads(open("$O/bench_g$1_$2_atomics$mode.json").read())
print("atomics=$mode G=$1 $2", d["ms_per_step"], d["roofline"]["frac"], d["result"], d.get("parity","")[:30])
PY
done
done
timeout 300 python tools/time_small.py c2:1e9 c3:1e9 c4:1e9 c4:125e6 c2:1e7 >> $O/small.log 2>&1; cat $O/small.log
tail -3 $O/bench.err
