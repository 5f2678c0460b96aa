export TMPDIR=/tmp EXON_HIP_INFLATE_PAR=1
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_inf/$n -o p --output-format csv -- python tools/time_inflate.py vcf 2e7 > /dev/null 2>&1
  f=$(find gpurun_out/pmc_inf/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'k_inflate' in k:
        acc[(k[:40],r['Counter_Name'])]+=float(r['Counter_Value']); n[(k[:40],r['Counter_Name'])]+=1
for k in sorted(acc): print(k, n[k], acc[k]/n[k])
PY
done
