#!/bin/bash
# where exon_hip_scan_next over a synthetic .vcf.gz spends its time (EXON_HIP_PIPE_TRACE lines + rocprofv3 kernel stats).  usage: [rows] [projection]
n=${1:-100000000}; pr=${2:-0}; d=$(mktemp -d /tmp/sn.XXXX)
tools/bin/gen_text vcf $n $d/s.vcf && tools/bin/bgzip $d/s.vcf $d/s.vcf.gz 6 && rm $d/s.vcf
EXON_HIP_PIPE_TRACE=1 tools/bin/time_scan_next $d/s.vcf.gz vcf 3 $pr 2>&1 | grep -E "export|setup|pass|teardown" | tail -9 | cut -c1-420
rm -rf $d
