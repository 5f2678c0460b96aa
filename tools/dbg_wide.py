#!/usr/bin/env python3
"""Debug helper for a symbol loop under test (EXON_HIP_INFLATE_FLAVOR): members of synthetic VCF / BAM / FASTQ one launch each,
then member by member: where the first differing byte of a bad member lies (and which DEFLATE block of the member it is in)."""
import os, subprocess, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exon_amd
BIN = os.path.join(ROOT, "tools", "bin")
ctx = exon_amd.Context(0)
kinds = sys.argv[1:] or ["vcf", "bam", "fastq"]
for kind in kinds:
    plain, comp = f"/tmp/dbgw.{kind}", f"/tmp/dbgw.{kind}.gz"
    subprocess.check_call([os.path.join(BIN, "gen_text"), kind, "60000", plain], stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(BIN, "bgzip"), plain, comp, "6"], stdout=subprocess.DEVNULL)
    raw = open(comp, "rb").read()
    blocks, n, consumed, out_bytes = exon_amd.bgzf_scan(raw)
    offs = []
    o = 0
    while o + 18 <= len(raw):
        bs = (raw[o + 16] | (raw[o + 17] << 8)) + 1
        offs.append((o, bs))
        o += bs
    bad = 0
    for i, (o, bs) in enumerate(offs):
        member = raw[o:o + bs]
        want = zlib.decompress(member, 31)
        try:
            got, _ = ctx.bgzf_inflate(member, verify_crc=False)
            got = got.tobytes()
            ok = got == want
            msg = "" if ok else "first difference at byte %d of %d" % (next((j for j in range(min(len(want), len(got))) if got[j] != want[j]), -1), len(want))
            if not ok:
                j = next((j for j in range(min(len(want), len(got))) if got[j] != want[j]), -1)
                msg += "; want %r got %r" % (want[max(0, j - 8):j + 24], got[max(0, j - 8):j + 24])
        except Exception as e:  # noqa: BLE001
            ok, msg = False, str(e)[-70:]
        if not ok:
            bad += 1
            if bad <= 6:
                print(f"{kind} member {i} ({bs} -> {len(want)} bytes): {msg}")
    print(f"{kind}: {bad} of {len(offs)} members differ")
# the reference's fixtures, whole files
import glob
for f in sorted(glob.glob(os.path.join(ROOT, "tests/golden/ref_fixtures/*.gz")) + glob.glob(os.path.join(ROOT, "tests/golden/ref_fixtures/*.bam")) + glob.glob(os.path.join(ROOT, "tests/golden/ref_fixtures/*.bcf"))):
    raw = open(f, "rb").read()
    if raw[:4] != b"\x1f\x8b\x08\x04":
        continue
    o, k = 0, 0
    while o + 18 <= len(raw):
        bs = (raw[o + 16] | (raw[o + 17] << 8)) + 1
        member = raw[o:o + bs]
        want = zlib.decompress(member, 31)
        try:
            got, _ = ctx.bgzf_inflate(member, verify_crc=False)
            got = got.tobytes()
            if got != want:
                j = next((j for j in range(min(len(want), len(got))) if got[j] != want[j]), -1)
                print(f"{os.path.basename(f)} member {k}: first difference at byte {j} of {len(want)}; want {want[max(0, j - 8):j + 24]!r} got {got[max(0, j - 8):j + 24]!r}")
        except Exception as e:  # noqa: BLE001
            print(f"{os.path.basename(f)} member {k}: {str(e)[-70:]}")
        o += bs
        k += 1
print("fixtures done")
