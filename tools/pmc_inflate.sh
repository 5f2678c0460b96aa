cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tools/bin/gen_text vcf 10000000 /tmp/p.vcf && tools/bin/bgzip /tmp/p.vcf /tmp/p.vcf.gz 6
cat > /tmp/inf_one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
raw = open("/tmp/p.vcf.gz","rb").read()
got, dt = ctx.bgzf_inflate(raw, verify_crc=False)
PY
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_inf -o run --output-format csv -- python /tmp/inf_one.py > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
fs = sorted(glob.glob("gpurun_out/pmc_inf/**/*counter_collection.csv", recursive=True))
agg = collections.defaultdict(float)
for f in fs:
    for r in csv.DictReader(open(f)):
        if "k_inflate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v/6558:.0f}/blk" for k, v in agg.items()})
PY
  rm -rf gpurun_out/pmc_inf
done
