#!/bin/bash
# BAM / BCF pipelines, same box: record chain walk with and without the line requests ahead of it (two library builds), and the
# BAM pipeline under both symbol loops.  tools/ab_chain.sh <out> <libA> <libB>
out=$1; a=$2; b=$3
mkdir -p $out
[ -f /tmp/e2e.bam ] || { tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6; }
[ -f /tmp/e2e.bcf ] || { tools/bin/gen_text bcf 50000000 /tmp/e2e.ubcf && tools/bin/bgzip /tmp/e2e.ubcf /tmp/e2e.bcf 6; }
cat /tmp/e2e.bam /tmp/e2e.bcf > /dev/null
for pass in 1 2; do
  for lib in $a $b; do
    for fl in hint 1; do
      for spec in "/tmp/e2e.bam bam" "/tmp/e2e.bcf bcf"; do
        echo "== pass $pass $(basename $lib) flavor=$fl $spec" >> $out/ab_chain.log
        if [ $fl = hint ]; then EXON_HIP_LIB=$lib python tools/time_pipeline_file.py $spec 6 >> $out/ab_chain.log 2>&1
        else EXON_HIP_LIB=$lib EXON_HIP_INFLATE_FLAVOR=$fl python tools/time_pipeline_file.py $spec 6 >> $out/ab_chain.log 2>&1; fi
      done
    done
  done
done
grep -E "^==|best" $out/ab_chain.log | paste - - | cut -c1-190
