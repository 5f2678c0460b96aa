#!/bin/bash
# round 3, GPU session 1: parity with the fused fold, then A/B of fold and tile shape on the launch-bound sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for f in 1 0; do
  EXON_HIP_FUSE_FOLD=$f timeout 300 python tools/time_small.py >> $O/ab.log 2>&1
done
for sh in 1 2; do
  EXON_HIP_SHAPE=$sh timeout 300 python tools/time_small.py c2:1e7 c2:2e7 c4:1e7 c3:1e8 c4:125e6 >> $O/ab.log 2>&1
done
cat $O/ab.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 3000 $O/bench_c4.json
