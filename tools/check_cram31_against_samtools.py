#!/usr/bin/env python3
"""Pins the CRAM 3.1 reader on htslib-written files -- for a machine that HAS samtools (this build image does not, which is why
HISTORY.md section 7j calls the rANS Nx16 decoder "unpinned against htslib").  No GPU needed: the host decoder is driven through
`exon_amd.Scan`.

    python tools/check_cram31_against_samtools.py input.bam|input.sam|input.cram [more files ...]

For every input and every htslib CRAM profile it writes a version-3.1 CRAM with samtools (`no_ref`: no FASTA needed), decodes it
with this library's host reader and compares flag, reference, start, end (start + reference span of the CIGAR - 1) and mapping
quality of every record with `samtools view` of the same file.  The `fast` / `normal` / `small` profiles put rANS Nx16 on the data
series this path reads (and the name tokeniser / fqzcomp on names and quality scores, whose blocks this reader never opens); the
`archive` profile may move series to the adaptive arithmetic coder, which this reader reports as unsupported by name -- that
outcome is printed as such and is not a mismatch.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyarrow as pa  # noqa: E402

import exon_amd  # noqa: E402

CONSUMES_REF = set("MDN=X")


def samtools_columns(path):
    out = subprocess.run(["samtools", "view", path], check=True, capture_output=True, text=True).stdout
    rows = []
    for line in out.splitlines():
        f = line.split("\t")
        flag, rname, pos, mapq, cigar = int(f[1]), f[2], int(f[3]), int(f[4]), f[5]
        span = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar) if op in CONSUMES_REF) if cigar != "*" else 0
        rows.append((flag, None if rname == "*" else rname, pos if pos > 0 else None, pos + span - 1 if pos > 0 else None,
                     None if mapq == 255 or flag & 4 else mapq))
    return rows


def product_columns(path):
    scan = exon_amd.Scan(path, "cram")
    names = scan.dictionary(2)
    rows = []
    for b in scan:
        b = pa.RecordBatch.from_struct_array(b) if isinstance(b, pa.StructArray) else b
        cols = [b.column(i).to_pylist() for i in range(5)]
        for flag, mapq, ref, start, end in zip(*cols):
            rows.append((flag, ref if ref is None or isinstance(ref, str) else names[ref], start, end, None if flag & 4 else mapq))
    scan.close()
    return rows


def main():
    if not shutil.which("samtools"):
        raise SystemExit("samtools is not on PATH: this check needs htslib to write CRAM 3.1")
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for src in sys.argv[1:]:
            for profile in ("fast", "normal", "small", "archive"):
                out = os.path.join(d, f"{os.path.basename(src)}.{profile}.cram")
                subprocess.run(["samtools", "view", "-O", f"cram,version=3.1,no_ref=1,{profile}", "-o", out, src], check=True)
                want = samtools_columns(out)
                try:
                    got = product_columns(out)
                except exon_amd.ExonHipError as e:
                    named = any(w in str(e) for w in ("arithmetic", "fqzcomp", "tokeniser"))
                    print(f"{src} [{profile}]: reader refused the file: {e}" + ("  (a codec it names as unsupported)" if named else "  <-- UNEXPECTED"))
                    bad += not named
                    continue
                if got == want:
                    print(f"{src} [{profile}]: {len(got)} records, all five columns equal samtools view")
                else:
                    first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
                    print(f"{src} [{profile}]: MISMATCH at record {first}: {got[first:first + 1]} vs {want[first:first + 1]} ({len(got)} / {len(want)} records)")
                    bad += 1
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
