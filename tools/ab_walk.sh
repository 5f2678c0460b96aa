#!/bin/bash
# reader-thread breakdown of the BGZF pipelines with / without the header warm pass (EXON_HIP_BGZF_WALK_WARM): tools/ab_walk.sh <out>
out=$1; mkdir -p $out
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
[ -f /tmp/e2e.fastq.gz ] || { tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6; }
[ -f /tmp/e2e.bam ] || { tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6; }
cat /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz /tmp/e2e.bam > /dev/null
g++ -O2 -pthread -Iinclude -o /tmp/time_bgzf_walk tools/time_bgzf_walk.cpp -Lexon_amd/lib -lexon_hip -Wl,-rpath,$PWD/exon_amd/lib
for w in 0 1; do echo "== walk only, WARM=$w" >> $out/ab_walk.log; EXON_HIP_BGZF_WALK_WARM=$w /tmp/time_bgzf_walk /tmp/e2e.vcf.gz >> $out/ab_walk.log 2>&1; done
for pass in 1 2; do
  for w in 0 1; do
    for spec in "/tmp/e2e.vcf.gz vcf" "/tmp/e2e.bam bam" "/tmp/e2e.fastq.gz fastq"; do
      echo "== pass $pass WARM=$w $spec" >> $out/ab_walk.log
      EXON_HIP_BGZF_WALK_WARM=$w python tools/time_pipeline_file.py $spec 6 >> $out/ab_walk.log 2>&1
    done
  done
done
for w in 0 1; do
  echo "== traced WARM=$w vcf" >> $out/ab_walk.log
  EXON_HIP_PIPE_TRACE=1 EXON_HIP_BGZF_WALK_WARM=$w python tools/time_pipeline_file.py /tmp/e2e.vcf.gz vcf 3 2>&1 | grep -E "teardown|loop|best" | tail -4 >> $out/ab_walk.log
done
grep -E "^==|best|teardown|ns per" $out/ab_walk.log
