#!/bin/bash
# round 3, GPU session 28: K5 path A = variant M in the product: parity (incl. the new steady-state non-ASCII cases), c5 bench, tune harness
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s28; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "k5 or c5 or qual or fastq or golden or fullsize" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --workload c5 2>/dev/null | tail -1 > $O/bench_c5.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/r3_s28/bench_c5.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])
PY
TUNE_K5_ONLY_IJ=1 timeout 300 tools/bin/tune_k5 1e8 100 2>&1 | grep -E "^K|^M|^R" | tail -12
