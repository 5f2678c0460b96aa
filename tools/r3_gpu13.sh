#!/bin/bash
# round 3, GPU session 13: final refresh at HEAD -- traced profiles of c2 / c3 / c6 / c4 (shapes changed), GROUP BY 64 keys,
# the full -m gpu suite, smoke, and an untraced default bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s13; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c4_untraced.json 2> $O/bench_c4.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_s13/bench_c4_untraced.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], json.dumps(d["extras"]["configs_at_stated_size"]))
PY
bash tools/refresh_profiles.sh r3 "c4 c2 c3 c6" > $O/refresh.log 2>&1; tail -8 $O/refresh.log
rocprofv3 --kernel-trace --stats -d $O/tmp_g64 -o g64 --output-format csv -- python bench.py --groups 64 --group-dist uniform --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_g64.json 2> /dev/null
cp $(find $O/tmp_g64 -name "*kernel_stats.csv" | head -1) $O/g64_kernel_stats.csv; rm -rf $O/tmp_g64
tail -1 $O/bench_g64.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('g64', d['ms_per_step'], d['roofline']['frac'])"
