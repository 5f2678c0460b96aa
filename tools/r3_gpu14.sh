#!/bin/bash
# round 3, GPU session 14: phases of the .vcf.gz pipeline (first run vs steady state), fuzz of the device decoders at HEAD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s14; mkdir -p $O
tools/bin/gen_text vcf 100000000 /tmp/e2e.vcf && tools/bin/bgzip /tmp/e2e.vcf /tmp/e2e.vcf.gz 6
cat > /tmp/vcfgz_trace.py <<PY
import sys, os, time
sys.path.insert(0, os.getcwd())
import exon_amd
ctx = exon_amd.Context(0)
for rep in range(4):
    t0 = time.perf_counter()
    scan = exon_amd.Scan("/tmp/e2e.vcf.gz", "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    t1 = time.perf_counter()
    st = plan.open(); rows = st.consume(scan); t2 = time.perf_counter(); st.finish(); st.close(); plan.close(); scan.close()
    print("run", rep, rows, "rows: open %.1f ms, consume %.1f ms, finish+close %.1f ms, total %.4f s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3, time.perf_counter() - t0), flush=True)
PY
EXON_HIP_PIPE_TRACE=1 python /tmp/vcfgz_trace.py > $O/vcfgz_trace.log 2>&1; cat $O/vcfgz_trace.log | grep -v amdgpu.ids
FUZZ_SEED=31 timeout 900 python tools/fuzz_gpu_decode.py 120 > $O/fuzz.log 2>&1; tail -12 $O/fuzz.log
FUZZ_SEED=32 timeout 900 python tools/fuzz_gpu_decode.py big 10 >> $O/fuzz.log 2>&1; tail -7 $O/fuzz.log
