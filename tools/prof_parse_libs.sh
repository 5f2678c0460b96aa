#!/bin/bash
# standalone VCF parse kernels (one resident slab) under several library builds: tools/prof_parse_libs.sh <out> "<lib> ..." [rows]
out=$1; libs=$2; rows=${3:-12000000}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
for lib in $libs; do
  echo "== $lib" >> $out/parse_libs.log
  [ $lib = default ] && unset EXON_HIP_LIB || export EXON_HIP_LIB=$lib
  python tools/time_gpu_parse.py $rows | tail -1 >> $out/parse_libs.log
  rocprofv3 --kernel-trace --stats -d $out/tmp -o p --output-format csv -- python tools/time_gpu_parse.py $rows > /dev/null 2>&1
  python3 - $(find $out/tmp -name p_kernel_stats.csv) >> $out/parse_libs.log <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:2]:
    print("   %-60s avg %9.1f us"%(r["Name"][:60],float(r["AverageNs"])/1e3))
PY
  rm -rf $out/tmp
done
cat $out/parse_libs.log
