"""Plain gzip (not BGZF) inflated on the GPU (exon_hip_gzip_stream_*, exon_amd/csrc/gzip_stream.hip) against zlib: byte-identical
output for every DEFLATE block type and strategy, back-references across chunk and slab boundaries, multi-member files, header-shaped
patterns planted in literal data, and loud failure on corrupt / truncated input.

The reference takes such files through `file_compression_type.convert_stream` (the `else` arm behind `is_bgzip_valid_header`,
exon-core/src/datasources/fastq/file_opener.rs:79-92); its fixtures test.fastq.gz / test.fasta.gz are plain gzip."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import exon_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")

pytestmark = pytest.mark.gpu


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8, wbits=31):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    return co.compress(data) + co.flush()


def vcf_like(n, seed=3):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        rows.append(f"{1 + i % 22}\t{1000 + i * 37}\trs{rng.integers(1, 10**7)}\t{'ACGT'[i % 4]}\t{'TGCA'[i % 4]}\t"
                    f"{rng.integers(0, 9999) / 10}\t{'PASS' if i % 5 else 'q10;s50'}\tAF={rng.random():.6f};DP={rng.integers(1, 500)}\n")
    return "".join(rows).encode()


def fastq_like(n, seed=4, L=100):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        seq = "".join("ACGT"[k] for k in rng.integers(0, 4, L))
        q = "".join(chr(33 + int(k)) for k in rng.integers(2, 41, L))
        out.append(f"@SRR1234567.{i} {i}/1\n{seq}\n+\n{q}\n")
    return "".join(out).encode()


def make(kind, rng):
    if kind == "text":
        return vcf_like(20000)
    if kind == "fastq":
        return fastq_like(4000)
    if kind == "random":
        return rng.integers(0, 256, 400_000, dtype=np.uint8).tobytes()
    if kind == "zeros":
        return bytes(300_000)
    if kind == "runs":
        return b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 700)) for _ in range(1500))
    if kind == "periods":
        return b"".join((bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)) * 300)[:int(rng.integers(100, 3000))] for _ in range(300))
    if kind == "short":
        return b"ACGT\n"
    if kind == "farlong":  # matches at distances close to 32 KiB
        blk = rng.integers(0, 256, 32000, dtype=np.uint8).tobytes()
        return blk + bytes(rng.integers(0, 256, 700, dtype=np.uint8)) + blk + blk[:5000] + blk
    raise KeyError(kind)


@pytest.fixture
def small_chunks(monkeypatch):
    monkeypatch.setenv("EXON_HIP_GZ_CHUNK_KB", "4")  # read at stream creation: many chunks out of small inputs


@pytest.mark.parametrize("name", ["fastq/test.fastq.gz", "fasta/test.fasta.gz"])
def test_reference_fixtures(ctx, name):
    raw = open(os.path.join(FX, name), "rb").read()
    assert ctx.gzip_inflate(raw) == gzip.decompress(raw)


@pytest.mark.parametrize("kind", ["text", "fastq", "random", "zeros", "runs", "periods", "short", "farlong"])
@pytest.mark.parametrize("level,strategy", [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_equals_zlib_every_strategy(ctx, small_chunks, kind, level, strategy):
    """7 strategies x 8 payloads, 4 KiB chunks (every payload but `short` spans many; back-references cross chunk boundaries
    everywhere in `text` / `periods` / `farlong`), one slab and 48 KiB slabs (blocks straddle slab ends, the tail is carried)."""
    data = make(kind, np.random.default_rng(5))
    raw = gz(data, level, strategy)
    got, st = ctx.gzip_inflate(raw, return_stats=True)
    assert got == data
    assert st["members"] == 1 and st["out_bytes"] == len(data)
    if len(raw) > 100_000:
        assert st["chunks"] > 8
    # (a stored block holds up to 65535 bytes: a slab must be larger than the largest block)
    got2, st2 = ctx.gzip_inflate(raw, slab_bytes=(48 if level else 96) << 10, out_cap=8 << 20, return_stats=True)
    assert got2 == data
    if len(raw) > 100_000:
        assert st2["calls"] > 1


def test_default_chunks_and_a_small_output_buffer(ctx):
    """64 KiB chunks (the default), and an output buffer that takes only part of a slab's chunks per call: the rest is decoded
    again by the next call (from the block boundary the accepted part ended at)."""
    data = vcf_like(120000, seed=9)
    raw = gz(data, 6)
    assert ctx.gzip_inflate(raw) == data
    got, st = ctx.gzip_inflate(raw, slab_bytes=1 << 20, out_cap=700_000, return_stats=True)
    assert got == data and st["calls"] > 5


def test_multi_member_files(ctx, small_chunks):
    """`cat a.gz b.gz c.gz`: members of different levels / strategies, an EMPTY member, header fields (FNAME, FEXTRA, FCOMMENT,
    FHCRC) -- each member's CRC-32 and ISIZE are checked; members end inside chunks and at slab ends."""
    rng = np.random.default_rng(11)
    parts = [vcf_like(3000, 1), b"", fastq_like(800, 2), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), bytes(50000), b"x"]
    params = [(6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
              (6, zlib.Z_RLE), (6, zlib.Z_FIXED)]
    raw = b"".join(gz(p, lv, s) for p, (lv, s) in zip(parts, params))

    def with_fields(payload):  # RFC 1952 header with every optional field
        body = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = body.compress(payload) + body.flush()
        head = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\0\xff" + struct.pack("<H", 5) + b"AB\x01\0z" + b"name.txt\0" + b"a comment\0"
        head += struct.pack("<H", zlib.crc32(head) & 0xFFFF)
        return head + cdata + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF)
    raw += with_fields(b"tail member\n" * 2000)
    want = b"".join(parts) + b"tail member\n" * 2000
    assert gzip.decompress(raw) == want
    got, st = ctx.gzip_inflate(raw, return_stats=True)
    assert got == want and st["members"] == 7
    got, st = ctx.gzip_inflate(raw, slab_bytes=80 << 10, out_cap=4 << 20, return_stats=True)  # (larger than the 65535-byte stored block)
    assert got == want and st["members"] == 7 and st["calls"] > 2


def test_many_tiny_members_in_one_chunk_hand_over(ctx):
    """more member ends in one chunk than its record holds (8): the call fails by name, the caller inflates on the host"""
    raw = b"".join(gz(b"r%d\n" % i) for i in range(40))
    with pytest.raises(exon_amd.ExonHipError) as e:
        ctx.gzip_inflate(raw)
    assert "members" in str(e.value)


def test_planted_block_headers_in_literal_data(ctx, small_chunks):
    """Header-shaped bit patterns where no block starts: the first bytes of REAL dynamic blocks (header + code lengths + some symbols)
    stored verbatim inside stored blocks and inside Huffman-only data.  The chunk search finds them, their chains die (or land
    somewhere the chunk in front did not stop), and the output is still zlib's."""
    real = zlib.compressobj(6, zlib.DEFLATED, -15)
    stream = real.compress(vcf_like(4000, 7)) + real.flush()
    assert (stream[0] & 7) == 4  # BFINAL = 0, BTYPE = 2
    rng = np.random.default_rng(13)
    bait = b"".join(stream[:int(rng.integers(60, 400))] + rng.integers(0, 256, int(rng.integers(10, 3000)), dtype=np.uint8).tobytes() for _ in range(120))
    for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_DEFAULT_STRATEGY)):
        raw = gz(bait, level, strategy) + gz(vcf_like(2000, 8))
        got, st = ctx.gzip_inflate(raw, return_stats=True)
        assert got == bait + vcf_like(2000, 8)
        if level == 0:
            assert st["repairs"] > 0  # stored data has no dynamic header to find: those chunks are decoded from the proven position


def test_corrupt_and_truncated_streams_fail(ctx, small_chunks):
    data = vcf_like(8000, 21)
    raw = bytearray(gz(data))
    want_crc = struct.unpack("<I", raw[-8:-4])[0]
    bad = bytearray(raw)
    bad[-8:-4] = struct.pack("<I", want_crc ^ 1)
    with pytest.raises(exon_amd.ExonHipError) as e:
        ctx.gzip_inflate(bytes(bad))
    assert "CRC" in str(e.value)
    bad = bytearray(raw)
    bad[-4:] = struct.pack("<I", len(data) + 1)
    with pytest.raises(exon_amd.ExonHipError) as e:
        ctx.gzip_inflate(bytes(bad))
    assert "ISIZE" in str(e.value)
    for cut in (len(raw) - 1, len(raw) - 9, len(raw) // 2, 30):
        with pytest.raises(exon_amd.ExonHipError):
            ctx.gzip_inflate(bytes(raw[:cut]))
    rng = np.random.default_rng(22)
    failures = 0
    for _ in range(40):  # a flipped bit anywhere in the DEFLATE data: an error, or (never seen) the same bytes
        bad = bytearray(raw)
        at = int(rng.integers(12, len(raw) - 8))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        try:
            got = ctx.gzip_inflate(bytes(bad))
            assert got == data
        except exon_amd.ExonHipError:
            failures += 1
    assert failures >= 39
    with pytest.raises(exon_amd.ExonHipError):
        ctx.gzip_inflate(b"\x1f\x8b\x07" + bytes(40))       # unknown method
    with pytest.raises(exon_amd.ExonHipError):
        ctx.gzip_inflate(bytes(raw) + b"trailing garbage")  # bytes behind the last member that are no member


def test_fuzz_3000_streams(ctx, small_chunks):
    """3000 random streams: payload kind, size, level, strategy, memLevel (block sizes), window bits (9..15), slab size."""
    rng = np.random.default_rng(2026)
    kinds = ["text", "fastq", "random", "runs", "periods"]
    base = {k: make(k, np.random.default_rng(31)) for k in kinds}
    n_multi = 0
    for it in range(3000):
        k = kinds[int(rng.integers(0, len(kinds)))]
        n = int(rng.integers(1, 60000))
        o = int(rng.integers(0, max(1, len(base[k]) - n)))
        data = base[k][o:o + n]
        level = int(rng.integers(0, 10))
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]
        mem = int(rng.integers(1, 10))
        wbits = 16 + int(rng.integers(9, 16))
        raw = gz(data, level, strategy, mem, wbits)
        if rng.random() < 0.2:
            raw += gz(base["text"][:int(rng.integers(0, 5000))], 6)
            data = data + base["text"][:len(gzip.decompress(raw)) - len(data)]
            n_multi += 1
        slab = None if rng.random() < 0.5 else int(rng.integers(8, 64)) << 10
        try:
            got = ctx.gzip_inflate(raw, slab_bytes=slab, out_cap=2 << 20)
        except exon_amd.ExonHipError as e:  # only legitimate hand-over: a block larger than the slab
            assert slab is not None and ("no progress" in str(e)), (it, str(e))
            got = ctx.gzip_inflate(raw, out_cap=2 << 20)
        assert got == data, (it, k, n, level, strategy, mem, wbits, slab)
    assert n_multi > 300


@pytest.mark.parametrize("flush", [zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])
def test_pigz_style_streams_with_empty_stored_blocks(ctx, small_chunks, flush):
    """pigz (and anything that uses Z_SYNC_FLUSH) ends every 128 KiB of input with an EMPTY STORED block; back-references go on
    across it.  The chunk search only looks for dynamic headers, so the chunk behind such a marker begins a few bytes late: the
    gaps are decoded all at once (one launch) and spliced in front of their chunks -- same bytes as zlib, one call."""
    data = vcf_like(60000, seed=17) + fastq_like(3000, seed=18)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    raw = b"".join(co.compress(data[i:i + 8192]) + co.flush(flush) for i in range(0, len(data), 8192)) + co.flush()
    assert gzip.decompress(raw) == data
    got, st = ctx.gzip_inflate(raw, return_stats=True)
    assert got == data and st["calls"] == 1 and st["repairs"] > 20
    got, st = ctx.gzip_inflate(raw, slab_bytes=64 << 10, out_cap=4 << 20, return_stats=True)
    assert got == data and st["calls"] > 3
