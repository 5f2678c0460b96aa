#!/bin/bash
# rocprofv3 PMC passes over the serial inflate kernel (one resident launch): tools/pmc_inflate.sh <out> [flavor] [kind rows]
# (separate --pmc passes with --kernel-trace only, as the MI355X guide prescribes)
out=$1; flavor=${2:-0}; kind=${3:-vcf}; rows=${4:-2e7}
export TMPDIR=/tmp EXON_TIME_INFLATE_NO_HOST=1 EXON_HIP_INFLATE_PAR=0 EXON_HIP_INFLATE_FLAVOR=$flavor
mkdir -p $out
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $out/tmp_$n -o p --output-format csv -- python tools/time_inflate.py $kind $rows > /dev/null 2>&1
  f=$(find $out/tmp_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out/pmc_${kind}_flavor$flavor.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r['Kernel_Name']
        if 'k_inflate' in k:
            acc[(k[:40],r['Counter_Name'])]+=float(r['Counter_Value']); n[(k[:40],r['Counter_Name'])]+=1
except Exception as e:
    print("no counters:", e)
for k in sorted(acc): print(k[1], "launches", n[k], "per launch", acc[k]/n[k])
PY
  rm -rf $out/tmp_$n
done
cat $out/pmc_${kind}_flavor$flavor.txt
