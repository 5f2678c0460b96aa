"""BGZF inflate on the GPU (exon_hip_bgzf_inflate) against zlib: byte-identical output for every DEFLATE block type,
CRC-32 verification, and loud failure on corrupt input."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import exon_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")


def bgzf_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, extra=b""):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    cdata = co.compress(data) + co.flush()
    xlen = 6 + len(extra)
    bsize = 12 + xlen + len(cdata) + 8
    assert bsize <= 65536
    head = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra + b"BC\x02\0" + struct.pack("<H", bsize - 1)
    return head + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def bgzf_file(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, chunk=65280):
    out = [bgzf_block(data[i:i + chunk], level, strategy) for i in range(0, len(data), chunk)]
    out.append(bgzf_block(b""))  # EOF marker
    return b"".join(out)


def vcf_like(n, seed=3):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        rows.append(f"{1 + i % 22}\t{1000 + i * 37}\trs{rng.integers(1, 10**7)}\t{'ACGT'[i % 4]}\t{'TGCA'[i % 4]}\t"
                    f"{rng.integers(0, 9999) / 10}\t{'PASS' if i % 5 else 'q10;s50'}\tAF={rng.random():.6f};DP={rng.integers(1, 500)}\n")
    return "".join(rows).encode()


def test_bgzf_scan_host():
    data = vcf_like(3000)
    f = bgzf_file(data, chunk=30000)
    blocks, n, consumed, out_bytes = exon_amd.bgzf_scan(f)
    assert n == (len(data) + 29999) // 30000 + 1 and consumed == len(f) and out_bytes == len(data)
    assert blocks[0].comp_offset == 18 and blocks[0].out_offset == 0 and blocks[1].out_offset == 30000
    assert blocks[n - 1].out_size == 0
    # a trailing partial block is left for the next call; an extra subfield before BC is skipped
    _, n2, consumed2, _ = exon_amd.bgzf_scan(f[:-5])
    assert n2 == n - 1 and consumed2 == len(f) - 28
    g = bgzf_block(b"hello world", extra=b"XY\x03\0abc")
    blocks, n, consumed, out_bytes = exon_amd.bgzf_scan(g)
    assert n == 1 and consumed == len(g) and out_bytes == 11 and blocks[0].comp_offset == 12 + 13
    with pytest.raises(exon_amd.ExonHipError):
        exon_amd.bgzf_scan(b"\x1f\x8b\x08\x00" + bytes(40))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["vcf/index.vcf.gz", "fastq/test_bgzip.fastq.gz", "bam/test.bam", "bcf/index.bcf"])
def test_gpu_inflate_reference_fixtures(ctx, name):
    raw = open(os.path.join(FX, name), "rb").read()
    want = gzip.decompress(raw)
    got, _ = ctx.bgzf_inflate(raw)
    assert got.tobytes() == want


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["text", "random", "zeros", "runs", "periods", "short", "farlong"])
@pytest.mark.parametrize("level,strategy", [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY),
                                            (6, zlib.Z_RLE)])
def test_gpu_inflate_equals_zlib(ctx, kind, level, strategy):
    rng = np.random.default_rng(5)
    if kind == "text":
        data = vcf_like(20000)
    elif kind == "random":
        data = rng.integers(0, 256, 400_000, dtype=np.uint8).tobytes()
    elif kind == "zeros":
        data = bytes(300_000)
    elif kind == "runs":  # overlapping copies with every small distance, long matches, sparse literals
        parts = []
        for d in range(1, 40):
            unit = rng.integers(65, 91, d, dtype=np.uint8).tobytes()
            parts.append(unit * (700 // d + 1))
        data = b"".join(parts) * 8
    elif kind == "periods":  # self-overlapping matches of every period 1..70 and length 3..600 behind 0..70 fresh literals: the
        parts = []           # wide loop folds a first symbol's bytes by whole periods; runs carried across rounds, periods at 63-65
        for i in range(3000):
            d, total = int(rng.integers(1, 71)), int(rng.integers(3, 601))
            unit = rng.integers(0, 256, d, dtype=np.uint8).tobytes()
            parts.append(rng.integers(0, 256, int(rng.integers(0, 71)), dtype=np.uint8).tobytes() + (unit * (total // d + 1))[:total])
        data = b"".join(parts)
    elif kind == "farlong":  # 258-byte matches from 3-5 KB back (beyond the 2 KiB ring): the far path with len > 64,
        unit = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()  # and short far ones at its seams
        data = b"".join(unit + rng.integers(0, 256, int(k), dtype=np.uint8).tobytes() for k in rng.integers(0, 2000, 60))
    else:
        data = b"".join(bytes([65 + i % 26]) * (i % 7) for i in range(200))
    chunk = 65280 if level else 60000
    if kind == "short":  # blocks of 0..40 bytes, including empty ones in the middle
        f = b"".join(bgzf_block(data[i:i + (i % 41)], level, strategy) for i in range(0, len(data), 41))
        want = b"".join(data[i:i + (i % 41)] for i in range(0, len(data), 41))
    else:
        f = bgzf_file(data, level, strategy, chunk)
        want = data
    got, _ = ctx.bgzf_inflate(f)
    assert got.tobytes() == want


@pytest.mark.gpu
def test_gpu_inflate_multi_deflate_blocks_per_member(ctx):
    """Z_FULL_FLUSH / Z_SYNC_FLUSH inside one member: several DEFLATE blocks (stored, empty, dynamic) back to back."""
    data = vcf_like(600)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = b""
    for i in range(0, len(data), 5000):
        cdata += co.compress(data[i:i + 5000]) + co.flush(zlib.Z_SYNC_FLUSH if (i // 5000) % 2 else zlib.Z_FULL_FLUSH)
    cdata += co.flush()
    bsize = 18 + len(cdata) + 8
    f = (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + cdata +
         struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    got, _ = ctx.bgzf_inflate(f)
    assert got.tobytes() == data


@pytest.mark.gpu
def test_gpu_inflate_reports_corruption(ctx):
    data = vcf_like(2000)
    f = bytearray(bgzf_file(data, chunk=20000))
    blocks, n, _, _ = exon_amd.bgzf_scan(bytes(f))
    bad_crc = bytearray(f)
    crc_at = blocks[1].comp_offset + blocks[1].comp_size
    bad_crc[crc_at] ^= 0x01
    with pytest.raises(exon_amd.ExonHipError, match="block 1: CRC-32 mismatch"):
        ctx.bgzf_inflate(bytes(bad_crc))
    got, _ = ctx.bgzf_inflate(bytes(bad_crc), verify_crc=False)  # the data itself is intact
    assert got.tobytes() == data
    flipped = bytearray(f)
    flipped[blocks[2].comp_offset + blocks[2].comp_size // 2] ^= 0x55
    with pytest.raises(exon_amd.ExonHipError, match="block 2"):
        ctx.bgzf_inflate(bytes(flipped))
    wrong_size = bytearray(f)
    at = blocks[0].comp_offset + blocks[0].comp_size + 4
    wrong_size[at:at + 4] = struct.pack("<I", blocks[0].out_size - 1)
    with pytest.raises(exon_amd.ExonHipError, match="block 0"):
        ctx.bgzf_inflate(bytes(wrong_size))


@pytest.mark.gpu
def test_gpu_inflate_fuzz_many_blocks_one_launch(ctx):
    """A few thousand blocks of random size, content class, level and strategy in ONE launch (one wavefront each): every
    byte equal to zlib's, CRCs verified on the device."""
    rng = np.random.default_rng(2026)
    text = vcf_like(4000, seed=9)
    blocks, want = [], []
    for i in range(3000):
        kind = i % 6
        size = int(rng.integers(0, 65281)) if i % 11 else int(rng.integers(0, 40))
        if kind == 0:
            off = int(rng.integers(0, max(1, len(text) - size)))
            data = text[off:off + size]
        elif kind == 1:
            data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
        elif kind == 2:
            data = rng.integers(0, 4, size, dtype=np.uint8).tobytes()          # 2-bit alphabet: very short codes
        elif kind == 3:
            data = (bytes(rng.integers(65, 70, 37, dtype=np.uint8)) * (size // 37 + 1))[:size]  # long far/near repeats
        elif kind == 4:
            data = bytes([int(rng.integers(0, 256))]) * size                    # one long run (distance 1)
        else:
            a = rng.integers(0, 256, size, dtype=np.uint8)
            a[rng.random(size) < 0.9] = 65                                       # sparse noise in a run
            data = a.tobytes()
        level = int(rng.choice([0, 1, 4, 6, 9]))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        if level == 0 and len(data) > 65000:
            data = data[:65000]
        blocks.append(bgzf_block(data, level, strategy))
        want.append(data)
    got, _ = ctx.bgzf_inflate(b"".join(blocks))
    assert got.tobytes() == b"".join(want)


@pytest.mark.gpu
def test_lane_parallel_block_decoder_in_a_fresh_process():
    """EXON_HIP_INFLATE_PAR=1 (speculative lane-parallel DEFLATE blocks, default off) is read once per process: this module's
    cases, the indexed region scans and the BAM pipeline rerun under it in a subprocess -- byte equality with zlib, CRC
    verification and corruption reports must not depend on which decoder ran."""
    import subprocess
    import sys
    env = dict(os.environ, EXON_HIP_INFLATE_PAR="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "--timeout", "300", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_inflate.py"), os.path.join(ROOT, "tests", "test_gpu_region_pushdown.py"),
                        os.path.join(ROOT, "tests", "test_gpu_bam_parse.py"),
                        "-k", "not fresh_process"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # mode 2: a share of every launch on the serial kernel, the rest on the parallel one, two streams joined by events
    env2 = dict(os.environ, EXON_HIP_INFLATE_PAR="2", EXON_HIP_INFLATE_PAR_SERIAL_SHARE="0.5")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "--timeout", "300", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_inflate.py"), "-k", "not fresh_process"], env=env2, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # and the parallel path did decode members (a silent fall-back to the serial loop would prove nothing)
    code = ("import exon_amd, ctypes, numpy as np, sys; sys.path.insert(0, %r); import test_gpu_inflate as t;"
            "ctx = exon_amd.Context(0); raw = t.bgzf_file(t.vcf_like(40000)); got, _ = ctx.bgzf_inflate(raw);"
            "st = (ctypes.c_uint32 * 32)(); ctx.lib.exon_hip_bgzf_inflate_par_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p];"
            "assert ctx.lib.exon_hip_bgzf_inflate_par_stats(None, st) == 0; print(st[0])") % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert int(r.stdout.strip().splitlines()[-1]) > 10


@pytest.mark.gpu
def test_every_symbol_loop_in_a_fresh_process():
    """EXON_HIP_INFLATE_FLAVOR picks the symbol loop of the serial kernels (read once per process): 3 = the wide loop (64 bit
    offsets per round; default since round 5), 1 = the software-pipelined vector-unit loop with deferred far copies, 0 = the scalar
    loop, 2 = loops 0 and 1 side by side.  Small
    launches take the lane-parallel decoder by default, so the serial kernel is forced (EXON_HIP_INFLATE_PAR=0) and this module's
    cases -- byte equality with zlib, CRC verification, corruption reports -- rerun under each loop."""
    import subprocess
    import sys
    for flavor in ("0", "1", "2", "3"):
        env = dict(os.environ, EXON_HIP_INFLATE_PAR="0", EXON_HIP_INFLATE_FLAVOR=flavor)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "--timeout", "300", "-p", "no:cacheprovider",
                            os.path.join(ROOT, "tests", "test_gpu_inflate.py"), "-k", "not fresh_process"], env=env, capture_output=True, text=True,
                           timeout=1500)
        assert r.returncode == 0, f"flavor {flavor}\n" + r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_gpu_inflate_compressed_slab_beyond_512_mib(ctx):
    """Members whose compressed bytes lie beyond 2^32 BITS of the slab (512 MiB): a symbol loop that keeps an absolute 32-bit bit
    index goes wrong there (round 5's wide loop did, on a 634 MB BAM slab).  One noisy text member repeated 12 000 times (~560 MB
    compressed); the device verifies every member's CRC-32, the host compares the first, one past the boundary, and the last."""
    rng = np.random.default_rng(11)
    a = np.frombuffer(vcf_like(1500, seed=4)[:65000], np.uint8).copy()
    noise = rng.random(a.size) < 0.55
    a[noise] = rng.integers(0, 256, int(noise.sum()), dtype=np.uint8)  # ~47 KB per member at level 6: matches AND long literal codes
    data = a.tobytes()
    blk = bgzf_block(data, 6)
    n = (560 << 20) // len(blk) + 1
    got, _ = ctx.bgzf_inflate(blk * n, verify_crc=True)
    assert got.size == n * len(data)
    for k in (0, (512 << 20) // len(blk) + 1, n - 1):
        assert got[k * len(data):(k + 1) * len(data)].tobytes() == data, k


@pytest.mark.gpu
@pytest.mark.parametrize("wbits,mem", [(15, 1), (9, 1), (9, 9), (12, 4)])
def test_gpu_inflate_many_small_deflate_blocks_and_small_windows(ctx, wbits, mem):
    """zlib's memLevel 1 cuts a member into DEFLATE blocks of 128 symbols (hundreds of table builds, end-of-block codes and
    symbol-loop entries per member); a 512-byte window (wbits 9) makes every match a near one.  Text, runs, sparse noise and
    literal-heavy bytes, every symbol loop's default path."""
    rng = np.random.default_rng(31)
    text = vcf_like(9000, seed=5)
    noisy = np.frombuffer(text, np.uint8).copy()
    m = rng.random(noisy.size) < 0.3
    noisy[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
    runs = b"".join(bytes([65 + i % 26]) * int(k) for i, k in enumerate(rng.integers(1, 400, 3000)))
    quals = rng.integers(33, 74, 600_000, dtype=np.uint8).tobytes()
    for data in (text, noisy.tobytes(), runs, quals):
        blocks = []
        for i in range(0, len(data), 65280):
            chunk = data[i:i + 65280]
            co = zlib.compressobj(6, zlib.DEFLATED, -wbits, mem)
            c = co.compress(chunk) + co.flush()
            if 18 + len(c) + 8 > 65536:   # incompressible under these settings: smaller members
                half = len(chunk) // 2
                for part in (chunk[:half], chunk[half:]):
                    co = zlib.compressobj(6, zlib.DEFLATED, -wbits, mem)
                    cc = co.compress(part) + co.flush()
                    blocks.append((part, cc))
            else:
                blocks.append((chunk, c))
        f = b"".join(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(c) + 8 - 1) + c +
                     struct.pack("<II", zlib.crc32(d) & 0xFFFFFFFF, len(d)) for d, c in blocks)
        got, _ = ctx.bgzf_inflate(f)
        assert got.tobytes() == b"".join(d for d, _ in blocks)
