#!/bin/bash
# round 3, GPU session 26: re-entry sanity at HEAD: the whole -m gpu suite, smoke, the default bench line, c5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s26; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > $O/bench_c4.json; cut -c1-600 $O/bench_c4.json
python bench.py --workload c5 2>/dev/null | tail -1 > $O/bench_c5.json; cut -c1-900 $O/bench_c5.json
