# inflate kernel time on resident compressed slabs of the three payload kinds (kernel only, 6144-block launches)
cd $GRAFT_REPO_ROOT
tools/bin/gen_text vcf 12000000 /tmp/p.vcf && tools/bin/bgzip /tmp/p.vcf /tmp/p.vcf.gz 6
tools/bin/gen_text fastq 1300000 /tmp/p.fq 150 0 && tools/bin/bgzip /tmp/p.fq /tmp/p.fq.gz 6
tools/bin/gen_text bam 2500000 /tmp/p.ubam 100 && tools/bin/bgzip /tmp/p.ubam /tmp/p.bam 6
cat > /tmp/inf_one.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C
import exon_amd
ctx = exon_amd.Context(0)
for f in ("/tmp/p.vcf.gz", "/tmp/p.fq.gz", "/tmp/p.bam"):
    raw = open(f, "rb").read()
    blocks, n, consumed, out_bytes = exon_amd.bgzf_scan(raw)
    nb = min(n, 6144)
    out_b = blocks[nb - 1].out_offset + blocks[nb - 1].out_size
    comp = np.frombuffer(raw, np.uint8)[:consumed]
    d_comp = ctx.to_device(np.concatenate([comp, np.zeros(4096 + (-len(comp)) % 4, np.uint8)]))
    d_out = ctx.empty(np.uint8, out_bytes + 64)
    bad = C.c_int32(-1)
    ts = []
    for rep in range(4):
        t = time.perf_counter()
        ctx._check(ctx.lib.exon_hip_bgzf_inflate(ctx.h, None, d_comp.ptr, blocks, nb, d_out.ptr, 0, C.byref(bad)))
        ts.append(time.perf_counter() - t)
    print(f"{f}: {nb} blocks, {out_b/1e6:.0f} MB out in {min(ts)*1e3:.2f} ms = {out_b/min(ts)/1e9:.1f} GB/s")
PY
python /tmp/inf_one.py
