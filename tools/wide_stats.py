#!/usr/bin/env python3
"""Where the wide inflate loop's clocks go (a library built with -DEXON_WIDE_STATS: tools/build_variant.sh stats -DEXON_WIDE_STATS,
EXON_HIP_LIB=exon_amd/lib/libexon_hip_stats.so): share of a member's clocks inside the hand-written rounds, and how often the rounds
hand back to the C++ glue, by reason.  usage: wide_stats.py {vcf|bam|fastq} ROWS"""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exon_amd  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind, n = sys.argv[1], int(float(sys.argv[2]))
plain, comp = f"/tmp/inf_bench.{kind}.{n}", f"/tmp/inf_bench.{kind}.{n}.6.gz"
if not (os.path.exists(plain) and os.path.exists(comp)):
    subprocess.check_call([os.path.join(ROOT, "tools/bin/gen_text"), kind, str(n), plain])
    subprocess.check_call([os.path.join(ROOT, "tools/bin/bgzip"), plain, comp, "6"])
raw = open(comp, "rb").read()
ctx = exon_amd.Context(0)
st = (ctypes.c_uint64 * 16)()
ctx.lib.exon_hip_bgzf_inflate_par_stats.restype = ctypes.c_int
ctx.lib.exon_hip_bgzf_inflate_par_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ctx.bgzf_inflate(raw, verify_crc=False)
ctx.lib.exon_hip_bgzf_inflate_par_stats(None, st)  # (reads and clears)
got, dt = ctx.bgzf_inflate(raw, verify_crc=False)
ctx.lib.exon_hip_bgzf_inflate_par_stats(None, st)
w = list(st)
members = max(w[8], 1)
names = ["unresolved symbol / end of block", "row drain for the caller", "?", "?", "bad distance", "slow round"]
print(f"{kind}: {members} members, {len(got) / dt / 1e9:.1f} GB/s; {w[1] / members:.0f} clocks per member, {100.0 * w[0] / max(w[1], 1):.1f} % inside the hand-written rounds")
for c in (0, 1, 4, 5):
    print(f"  hand-backs per member, {names[c]}: {w[2 + c] / members:.1f}")
