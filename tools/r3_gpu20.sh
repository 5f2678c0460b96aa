#!/bin/bash
# round 3, GPU session 20: timeline of the .vcf.gz pipeline (per-dispatch start/end of inflate and parse kernels)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s20; mkdir -p $O
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat /tmp/e2e.vcf.gz > /dev/null
EXON_HIP_PIPE_TRACE=1 rocprofv3 --kernel-trace -d $O/tmp -o tl --output-format csv -- python tools/trace_vcfgz.py /tmp/e2e.vcf.gz 3 > $O/run.log 2>&1
f=$(find $O/tmp -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# last third = the third run (steady)
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    print("%10.3f %10.3f %8.3f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name))
PY
grep -E "^run|setup" $O/run.log
wc -l $O/timeline.txt; rm -rf $O/tmp
