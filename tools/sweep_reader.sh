#!/bin/bash
# read-pool width x staging piece size of the BGZF pipelines' reader thread: tools/sweep_reader.sh <out> "threads..." "piece MB..."
out=$1; threads=${2:-"8 12 16"}; pieces=${3:-"8 16"}
mkdir -p $out
[ -f /tmp/e2e.vcf.gz ] || { tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6; }
[ -f /tmp/e2e.fastq.gz ] || { tools/bin/gen_text fastq 20000000 /tmp/e2e.fastq && tools/bin/bgzip /tmp/e2e.fastq /tmp/e2e.fastq.gz 6; }
[ -f /tmp/e2e.bam ] || { tools/bin/gen_text bam 20000000 /tmp/e2e.ubam && tools/bin/bgzip /tmp/e2e.ubam /tmp/e2e.bam 6; }
cat /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz /tmp/e2e.bam > /dev/null
ls -l /tmp/e2e.vcf.gz /tmp/e2e.fastq.gz /tmp/e2e.bam >> $out/sweep_reader.log
for pass in 1 2; do
  for t in $threads; do
    for p in $pieces; do
      for spec in "/tmp/e2e.fastq.gz fastq" "/tmp/e2e.bam bam" "/tmp/e2e.vcf.gz vcf"; do
        echo "== pass $pass threads=$t piece=$p MB $spec" >> $out/sweep_reader.log
        EXON_HIP_READ_THREADS=$t EXON_HIP_RING_PIECE_MB=$p python tools/time_pipeline_file.py $spec 6 >> $out/sweep_reader.log 2>&1
      done
    done
  done
done
grep -E "^==|best|e2e" $out/sweep_reader.log
