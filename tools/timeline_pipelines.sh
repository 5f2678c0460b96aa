#!/bin/bash
# where the inflate stream idles in the three BGZF pipelines (rocprofv3 --kernel-trace + tools/timeline_gaps.py): tools/timeline_pipelines.sh <out>
out=$1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p $out
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
[ -f /tmp/e2e.fastq.gz ] || { tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6; }
[ -f /tmp/e2e.bam ] || { tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6; }
cat /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz /tmp/e2e.bam > /dev/null
for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.bam bam" "/tmp/e2e.fastq.gz fastq"; do
  set -- $spec
  rocprofv3 --kernel-trace -d $out/tmp_$2 -o $2 --output-format csv -- python tools/time_pipeline_file.py $1 $2 3 > $out/$2_traced.log 2>&1
  echo "== $2" >> $out/timeline.log
  tail -1 $out/$2_traced.log >> $out/timeline.log
  python tools/timeline_gaps.py $(find $out/tmp_$2 -name "*kernel_trace.csv" | head -1) >> $out/timeline.log 2>&1
  cp $(find $out/tmp_$2 -name "*kernel_trace.csv" | head -1) $out/$2_kernel_trace.csv; rm -rf $out/tmp_$2
done
cat $out/timeline.log
