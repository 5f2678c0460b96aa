# PMC counters of k_parse_lines on one resident slab of VCF text (developer tool)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_parse -o run --output-format csv -- python tools/time_gpu_parse.py 3000000 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
fs = sorted(glob.glob("gpurun_out/pmc_parse/**/*counter_collection.csv", recursive=True))
agg = collections.defaultdict(float); n = 0
for f in fs:
    for r in csv.DictReader(open(f)):
        if "k_parse_lines" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v/4/3e6:.1f}/line" for k, v in agg.items()})
PY
  rm -rf gpurun_out/pmc_parse
done
