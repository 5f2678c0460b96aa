#!/bin/bash
# round 3, GPU session 30: K5 path A for any uniform read length: parity, timing per length, c5 bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s30; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "k5 or c5 or qual or fastq or golden or fullsize" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
python tools/time_k5_lengths.py 2>&1 | grep -v amdgpu.ids | tee $O/k5_lengths.log
python bench.py --workload c5 2>/dev/null | tail -1 > $O/bench_c5.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/r3_s30/bench_c5.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])
PY
