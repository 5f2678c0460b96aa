#!/bin/bash
# round 3, GPU session 21: inflate stream holds kernels only (table on the copy stream, status zero-copy): timeline + end to end
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s21; mkdir -p $O
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat /tmp/e2e.vcf.gz > /dev/null
EXON_HIP_PIPE_TRACE=1 python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 5 2>&1 | grep -v amdgpu.ids > $O/trace.log
grep -E "^run|setup|teardown" $O/trace.log | tail -9
EXON_HIP_PIPE_TRACE=1 rocprofv3 --kernel-trace -d $O/tmp -o tl --output-format csv -- python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 3 > $O/run.log 2>&1
f=$(find $O/tmp -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = r["Kernel_Name"]
    name = name[name.find("k_"):][:24] if "k_" in name else name.split("(")[0][-28:]
    print("%10.3f %10.3f %8.3f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name))
PY
rm -rf $O/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "pipeline or scan or vcf or bam or bcf or fastq or sam or region or cram or inflate or bgzf" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
