d=$(mktemp -d /tmp/sn.XXXX)
tools/bin/gen_text vcf 30000000 $d/s.vcf && tools/bin/bgzip $d/s.vcf $d/s.vcf.gz 6 && rm $d/s.vcf
export TMPDIR=/tmp
EXON_HIP_PIPE_TRACE=1 tools/bin/time_scan_next $d/s.vcf.gz vcf 3 7 2>&1 | grep -E "export|setup|pass" | tail -10
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/snprof -o t --output-format csv -- /root/repo/tools/bin/time_scan_next $d/s.vcf.gz vcf 3 7 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/snprof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
rm -rf $d /tmp/snprof
