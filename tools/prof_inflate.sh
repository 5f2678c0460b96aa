# inflate kernel time vs number of blocks (resident compressed slab), VCF text level 6
cd $GRAFT_REPO_ROOT
tools/bin/gen_text vcf 10000000 /tmp/p.vcf && tools/bin/bgzip /tmp/p.vcf /tmp/p.vcf.gz 6
cat > /tmp/inf_one.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C
import exon_amd
ctx = exon_amd.Context(0)
raw = open("/tmp/p.vcf.gz","rb").read()
blocks, n, consumed, out_bytes = exon_amd.bgzf_scan(raw)
comp = np.frombuffer(raw, np.uint8)[:consumed]
d_comp = ctx.to_device(np.concatenate([comp, np.zeros(4096 + (-len(comp)) % 4, np.uint8)]))
d_out = ctx.empty(np.uint8, out_bytes + 64)
bad = C.c_int32(-1)
for nb in (1, 2048, 6558):
    ts = []
    for rep in range(3):
        t = time.perf_counter()
        ctx._check(ctx.lib.exon_hip_bgzf_inflate(ctx.h, None, d_comp.ptr, blocks, nb, d_out.ptr, 0, C.byref(bad)))
        ts.append(time.perf_counter() - t)
    print(f"{nb} blocks: {min(ts)*1e3:.3f} ms")
PY
python /tmp/inf_one.py
