#!/bin/bash
# rocprofv3 PMC passes over k_gz_decode (plain gzip on the GPU, tools/time_gz_decode.py): tools/pmc_gz.sh <out> [kind rows]
# (separate --pmc passes with --kernel-trace only, as the MI355X guide prescribes)
out=$1; kind=${2:-vcf}; rows=${3:-20000000}
export TMPDIR=/tmp
mkdir -p $out
python tools/time_gz_decode.py $kind $rows 256 1 > $out/pmc_gz_$kind.txt 2>&1   # (makes the file; prints the untraced rate)
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OLDPWD/$out/tmp_$n -o p --output-format csv -- python $OLDPWD/tools/time_gz_decode.py $kind $rows 256 2 > /dev/null 2>&1)
  f=$(find $out/tmp_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out/pmc_gz_$kind.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r['Kernel_Name']
        if 'k_gz_decode' in k and float(r['Grid_Size']) > 64*64:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
except Exception as e:
    print("no counters:", e)
for k in sorted(acc): print(k, "launches", n[k], "per launch", acc[k]/n[k])
PY
  rm -rf $out/tmp_$n
done
cat $out/pmc_gz_$kind.txt
