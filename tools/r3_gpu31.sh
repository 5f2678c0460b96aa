#!/bin/bash
# round 3, GPU session 31: config 5 at other read lengths through bench.py (parity gate included), batch-size question, full -m gpu suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s31; mkdir -p $O
for L in 100 151 101 250; do
  EXON_BENCH_C5_L=$L python bench.py --workload c5 --rows 6e8 --steps 8 2>/dev/null | tail -1 > $O/bench_c5_L$L.json
  python - $O/bench_c5_L$L.json $L <<'PY'
import json,sys; d=json.load(open(sys.argv[1])); print("L", sys.argv[2], d['ms_per_step'], d['roofline']['frac'], d.get('parity'))
PY
done
EXON_BENCH_C5_BATCH=21474836 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch 21474836', d['ms_per_step'], d['roofline']['frac'])"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
