#!/bin/bash
# round 3, GPU session 32: K5 path A on memory-aligned rows (any base alignment): parity, bench at several read lengths and batch sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s32; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "k5 or c5 or qual or fastq or golden or fullsize" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
for L in 100 151 101 250; do
  EXON_BENCH_C5_L=$L python bench.py --workload c5 --rows 6e8 --steps 8 2>/dev/null | tail -1 > $O/bench_c5_L$L.json
  python - $O/bench_c5_L$L.json $L <<'PY'
import json,sys; d=json.load(open(sys.argv[1])); print("L", sys.argv[2], d['ms_per_step'], d['roofline']['frac'], d.get('parity'))
PY
done
EXON_BENCH_C5_BATCH=21474836 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch 21474836', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['roofline']['frac'])"
