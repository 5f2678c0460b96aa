#!/usr/bin/env python3
"""Research spike driver: tools/simt_inflate.hip (one BGZF block per lane) against inflate.hip on the same file."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")
kind = sys.argv[1] if len(sys.argv) > 1 else "vcf"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
subprocess.check_call([os.path.join(BIN, "gen_text"), kind, str(n), "/tmp/simt.txt"] + (["100"] if kind == "bam" else []))
subprocess.check_call([os.path.join(BIN, "bgzip"), "/tmp/simt.txt", "/tmp/simt.gz", "6"])
raw = open("/tmp/simt.gz", "rb").read()
lib = C.CDLL(os.path.join(BIN, "libsimt_inflate.so"))
ctx = exon_amd.Context(0)
blocks, nb, consumed, out_bytes = exon_amd.bgzf_scan(raw)
print(f"{kind}: {len(raw) / 1e6:.0f} MB -> {out_bytes / 1e6:.0f} MB, {nb} blocks")
comp = np.frombuffer(raw, np.uint8)[:consumed]
d_comp = ctx.to_device(np.concatenate([comp, np.zeros(4096, np.uint8)]))
d_blocks = ctx.to_device(np.frombuffer(bytes(blocks)[:nb * 24], np.uint8).copy())
d_out = ctx.empty(np.uint8, out_bytes + 4096)
nb_pad = (nb + 63) // 64 * 64
d_scr = ctx.empty(np.uint8, nb_pad * lib.simt_scratch_bytes_per_block() + 64)
d_st = ctx.empty(np.int32, nb + 16)
ms = C.c_float()
lib.simt_inflate.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(C.c_float)]
for rep in range(2):
    rc = lib.simt_inflate(d_comp.ptr, d_blocks.ptr, nb, d_out.ptr, d_scr.ptr, d_st.ptr, 1, C.byref(ms))
    print(f"simt, tokenizer + literal stores only (output wrong): rc {rc}, {ms.value:.2f} ms = {out_bytes / ms.value / 1e6:.1f} GB/s-equivalent")
for rep in range(3):
    rc = lib.simt_inflate(d_comp.ptr, d_blocks.ptr, nb, d_out.ptr, d_scr.ptr, d_st.ptr, 0, C.byref(ms))
    print(f"simt: rc {rc}, {ms.value:.2f} ms = {out_bytes / ms.value / 1e6:.1f} GB/s out")
st = d_st.to_host(nb)
print("blocks handed back (status 100):", int((st != 0).sum()), "first codes:", st[st != 0][:8])
got = d_out.to_host(out_bytes)
want = np.fromfile("/tmp/simt.txt", np.uint8)
print("equal to the original:", bool(np.array_equal(got, want[:out_bytes])))
t = []
for rep in range(3):
    g2, dt = ctx.bgzf_inflate(raw, verify_crc=False)
    t.append(dt)
print(f"inflate.hip (one launch, all blocks): {min(t) * 1e3:.2f} ms = {out_bytes / min(t) / 1e9:.1f} GB/s out")
