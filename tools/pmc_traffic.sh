# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the dominant kernel of each bench workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for spec in "c2 1e9" "c3 1e9" "c6 1e9" "c5 2e8"; do
  set -- $spec
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmc_$( [ $ctr = FETCH_SIZE ] && echo fetch || echo write )
    rm -rf $d/tmp_$1
    rocprofv3 --pmc $ctr --kernel-trace -d $d/tmp_$1 -o $1 --output-format csv -- python bench.py --workload $1 --rows $2 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    f=$(find $d/tmp_$1 -name "*counter_collection.csv" | head -1)
    cp "$f" $d/$1_counter_collection.csv
    rm -rf $d/tmp_$1
  done
  python tools/collect_profiles.py r1 $1 $2 | tail -1
done
