#!/bin/bash
# round 3, GPU session 4: full -m gpu suite (ABI 3: Int32 INFO, 16 keys, goldens), bench line, PMC for group paths
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 1500 $O/bench_c4.json
