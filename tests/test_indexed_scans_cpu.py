"""CPU: indexed scans over files whose chunks start and end inside BGZF blocks (indexes written by tests/bgzf_index_writer.py):
the host planner (host/bgzf_index.h) + random-access block reader return exactly the records a brute-force filter keeps, and
the hardened BGZF block reader rejects the malformed blocks the advisor listed."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import exon_amd
from bgzf_index_writer import bgzf_blocks, sorted_bam, write_bai, write_tabix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")


def count_rows(path, fmt, **kw):
    s = exon_amd.Scan(path, fmt, **kw)
    n = sum(len(b) for b in s)
    chunks = s.index_chunks()
    s.close()
    return n, chunks


def test_host_indexed_vcf_equals_brute_force(tmp_path):
    n = 200_000
    path = tmp_path / "s.vcf"
    subprocess.check_call([GEN, "vcf", str(n), str(path)])
    gz = tmp_path / "s.vcf.gz"
    subprocess.check_call([BGZIP, str(path), str(gz), "6"])
    assert write_tabix(gz) == n and len(bgzf_blocks(open(gz, "rb").read())) > 50
    for region, want in [("1:50000-60000", 10_001), ("1:1-3", 3), (f"1:{n}-{n + 9}", 1), ("1", n), ("1:150001", 50_000),
                         ("1:900000-900001", 0), ("zz", 0)]:
        got, chunks = count_rows(gz, "vcf", region=region, use_index=True)
        assert got == want and (chunks >= 1 or want == 0), region
        assert count_rows(gz, "vcf", region=region)[0] == want  # full scan + per-record filter


def test_host_indexed_bam_equals_brute_force(tmp_path):
    rng = np.random.default_rng(3)
    n = 60_000
    ub = tmp_path / "s.ubam"
    rows = np.array(sorted_bam(ub, n, rng), np.int64)
    bam = tmp_path / "s.bam"
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    assert write_bai(bam) == n - n // 50
    for region, (rid, a, b) in [("chr2:10000000-20000000", (1, 10_000_000, 20_000_000)), ("chr1:1-5000", (0, 1, 5000)),
                                ("chr3", (2, 1, 2**62))]:
        want = int(((rows[:, 0] == rid) & (rows[:, 1] <= b) & (rows[:, 2] >= a)).sum())
        assert count_rows(bam, "bam", region=region, use_index=True)[0] == want
        assert count_rows(bam, "bam", region=region)[0] == want


def _block(payload, extra=None, bsize_field=None, isize=None, crc=None):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    data = comp.compress(payload) + comp.flush()
    extra = struct.pack("<2sHH", b"BC", 2, 0) if extra is None else extra
    total = 12 + len(extra) + len(data) + 8
    if b"BC" in extra:
        i = extra.index(b"BC")
        extra = extra[:i + 4] + struct.pack("<H", (total - 1) if bsize_field is None else bsize_field) + extra[i + 6:]
    head = b"\x1f\x8b\x08\x04" + bytes(6) + struct.pack("<H", len(extra)) + extra
    tail = struct.pack("<II", zlib.crc32(payload) if crc is None else crc, len(payload) if isize is None else isize)
    return head + data + tail


VCF = (b"##fileformat=VCFv4.3\n##contig=<ID=1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" +
       b"".join(b"1\t%d\t.\tA\tC\t1\tPASS\t.\n" % (i + 1) for i in range(2000)))


def _variant(what, payload=VCF):
    if what == "ok":
        return _block(payload)
    if what == "bc_not_first":
        return _block(payload, extra=struct.pack("<2sH4s", b"XY", 4, b"abcd") + struct.pack("<2sHH", b"BC", 2, 0))
    if what == "bsize_too_small":
        return _block(payload, bsize_field=10)
    if what == "xlen_huge":
        b = bytearray(_block(payload))
        b[10:12] = struct.pack("<H", 60000)
        return bytes(b)
    if what == "isize_huge":
        return _block(payload, isize=0xFFFFFFF0)
    if what == "crc":
        return _block(payload, crc=12345)
    raise ValueError(what)


BAD = ["bsize_too_small", "xlen_huge", "isize_huge", "crc"]


@pytest.mark.parametrize("what", ["ok", "bc_not_first"] + BAD)
def test_host_bgzf_readers_validate_every_length(tmp_path, what):
    """BSIZE < header + trailer, XLEN beyond the block, ISIZE up to 4 GiB and a wrong CRC-32 must be
    rejected by the host BGZF readers, and a BC subfield that is not the first one must be found.  Three readers: the
    sequential zlib one (small files), the random-access one (indexed scans), the block-parallel one (files >= 8 MiB)."""
    eof = _block(b"")
    # 1. sequential reader (zlib decodes gzip members and never looks at BSIZE: a wrong BSIZE alone is harmless there)
    p = tmp_path / f"{what}.vcf.gz"
    open(p, "wb").write(_variant(what) + eof)
    if what in ("ok", "bc_not_first", "bsize_too_small"):
        assert count_rows(p, "vcf")[0] == 2000
    else:
        with pytest.raises(exon_amd.ExonHipError):
            count_rows(p, "vcf")
    # 2. random-access reader: the index comes from the well-formed twin (same block offsets), then the block is swapped in
    q = tmp_path / f"idx_{what}.vcf.gz"
    good = _variant("bc_not_first" if what == "bc_not_first" else "ok") + eof
    open(q, "wb").write(good)
    assert write_tabix(q) == 2000
    bad = _variant(what) + eof
    assert len(bad) == len(good)
    open(q, "wb").write(bad)
    if what in ("ok", "bc_not_first"):
        assert count_rows(q, "vcf", region="1:5-14", use_index=True) == (10, 1)
    else:
        with pytest.raises(exon_amd.ExonHipError):
            count_rows(q, "vcf", region="1:5-14", use_index=True)


@pytest.mark.parametrize("what", ["ok"] + BAD)
def test_block_parallel_reader_validates_every_length(tmp_path, what):
    """A 9 MiB BGZF file (the block-parallel host reader takes over at 8 MiB) with one malformed block in the middle."""
    rng = np.random.default_rng(8)
    head = b"##fileformat=VCFv4.3\n##contig=<ID=1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    blocks, row, i = [_block(head)], 0, 0
    while sum(len(b) for b in blocks) < (9 << 20):
        lines = []
        for _ in range(400):  # incompressible ids keep the file big without millions of rows
            row += 1
            lines.append(b"1\t%d\t%s\tA\tC\t1\tPASS\t.\n" % (row, rng.bytes(48).hex().encode()))
        blocks.append(_variant(what if i == 100 else "ok", b"".join(lines)))
        i += 1
    p = tmp_path / "big.vcf.gz"
    open(p, "wb").write(b"".join(blocks) + _block(b""))
    if what == "ok":
        assert count_rows(p, "vcf")[0] == row
    else:
        with pytest.raises(exon_amd.ExonHipError):
            count_rows(p, "vcf")


def _reference_indexed_stream_rows(gz, region_name, a, b, batch=8192):
    """What IndexedAsyncBatchStream::read_batch returns, restated from exon-vcf/src/indexed_async_batch_stream.rs:118-166 over
    the records of every index chunk (one stream per chunk, indexed_bgzf_file.rs:145-152): per batch, records until `batch` of
    them hit the region, then up to `batch` further records UNFILTERED.  Independent of the product: the chunks come from
    exon_hip_index_query, the record -> virtual-offset map from this test's own BGZF walk."""
    from bgzf_index_writer import VirtualOffsets
    blocks = bgzf_blocks(open(gz, "rb").read())
    vo = VirtualOffsets(blocks)
    text = b"".join(d for _, _, d in blocks)
    recs, u = [], 0
    while u < len(text):
        e = text.find(b"\n", u) + 1
        if text[u:u + 1] != b"#":
            f = text[u:e].split(b"\t", 3)
            recs.append((vo.at(u), f[0].decode(), int(f[1])))
        u = e
    total = 0
    for c0, c1 in exon_amd.index_query(str(gz) + ".tbi", region=f"{region_name}:{a}-{b}"):
        stream = [r for r in recs if c0 <= r[0] < c1]
        i = 0
        while i < len(stream):
            hits = 0
            while hits < batch and i < len(stream):            # first loop: filtered
                _, ch, pos = stream[i]
                i += 1
                if ch == region_name and a <= pos <= b:
                    hits += 1
                    total += 1
            if hits == batch:                                    # second loop: unfiltered (it reads nothing at a chunk's end)
                n = min(batch, len(stream) - i)
                total += n
                i += n
    return total


def test_indexed_vcf_unfiltered_tail_of_the_reference_is_a_switch(tmp_path, monkeypatch):
    """More than 8192 hits in one index chunk.  Default: every record is tested (what vcf_region_filter documents) -> the hits.
    EXON_HIP_REFERENCE_QUIRKS=1: the reference's stream as written appends up to batch_size UNFILTERED records after a full
    batch of hits (exon-vcf/src/indexed_async_batch_stream.rs:143-154) -> more rows than hits.  Both are stated here, so
    "identical results on the same inputs" is a switch, not a footnote."""
    n = 60_000
    path = tmp_path / "s.vcf"
    subprocess.check_call([GEN, "vcf", str(n), str(path)])          # contig 1, POS = 1 .. n
    gz = tmp_path / "s.vcf.gz"
    subprocess.check_call([BGZIP, str(path), str(gz), "6"])
    assert write_tabix(gz) == n
    a, b = 1, 10_000
    monkeypatch.delenv("EXON_HIP_REFERENCE_QUIRKS", raising=False)
    assert count_rows(gz, "vcf", region=f"1:{a}-{b}", use_index=True)[0] == 10_000
    want_quirk = _reference_indexed_stream_rows(gz, "1", a, b)
    assert want_quirk > 10_000                                       # the chunk runs past the region: unfiltered records follow
    monkeypatch.setenv("EXON_HIP_REFERENCE_QUIRKS", "1")
    got, chunks = count_rows(gz, "vcf", region=f"1:{a}-{b}", use_index=True)
    assert got == want_quirk and chunks >= 1
    s = exon_amd.Scan(gz, "vcf", region=f"1:{a}-{b}", use_index=True)
    pos = np.concatenate([bt.field(1).to_numpy(zero_copy_only=False) for bt in s])
    s.close()
    assert int((pos <= b).sum()) == 10_000 and int((pos > b).sum()) == want_quirk - 10_000 and pos.max() <= 8192 * 2 + 8192
    # fewer than batch_size hits per chunk: the two behaviours coincide (every reference test of this path sits here)
    assert count_rows(gz, "vcf", region="1:50000-55000", use_index=True)[0] == 5001 == _reference_indexed_stream_rows(gz, "1", 50_000, 55_000)
    # a scan without the index is the plain filtered stream in both modes
    assert count_rows(gz, "vcf", region=f"1:{a}-{b}")[0] == 10_000
