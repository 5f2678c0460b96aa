#!/bin/bash
# LDS / VALU counters of the config-5 kernel at HEAD: two rocprofv3 --pmc passes of `bench.py --workload c5`
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_k5
for grp in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-48)
  rm -rf gpurun_out/pmc_k5/$n
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_k5/$n -o p --output-format csv -- python bench.py --workload c5 --rows 2e8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  f=$(find gpurun_out/pmc_k5/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'k5_main' in k:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(acc): print("k5_main", k, "launches", n[k], "per_launch", acc[k]/n[k])
PY
done
