"""Plain-gzip (non-BGZF) FASTQ / VCF / SAM files through exon_hip_stream_consume_scan: the compressed bytes are shipped as they are,
inflated on the GPU (gzip_stream.hip) and parsed there -- `decoded_on_gpu` AND `inflated` = 1 -- and the answers equal the BGZF twin's,
the plain-text twin's and the host decoders'.  The reference reads such files through the `else` arm of its openers
(exon-core/src/datasources/fastq/file_opener.rs:79-92); its own fixture test.fastq.gz is one."""
import gzip
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest

import exon_amd
from oracle import decode
from oracle_expect import k4_expected

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")

pytestmark = pytest.mark.gpu


def plain_gzip(src, dst, level=6, members=1):
    """`gzip -c`: ONE member (or `members` concatenated ones), no BGZF extra field"""
    data = open(src, "rb").read()
    step = (len(data) + members - 1) // members
    with open(dst, "wb") as f:
        for i in range(0, len(data), step):
            co = zlib.compressobj(level, zlib.DEFLATED, 31)
            f.write(co.compress(data[i:i + step]) + co.flush())


def hist(ctx, path, gpu_parse, lmax=256, want_gpu=None, compression=None):
    scan = exon_amd.Scan(str(path), "fastq", gpu_parse=gpu_parse, compression=compression)
    plan = ctx.plan_qual_pos_hist(lmax, columns=(3,))
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    st.close()
    plan.close()
    flags = scan.decoded_on_gpu()
    scan.close()
    if want_gpu is not None:
        assert tuple(bool(x) for x in flags) == want_gpu, flags
    return rows, np.array(counts)


def k4(ctx, path, gpu_parse, want_gpu=None, info_field="AF", thr=0.01, fmt="vcf"):
    scan = exon_amd.Scan(str(path), fmt, info_field=info_field, gpu_parse=gpu_parse)
    plan = ctx.plan_cmp_avg_by_group(">", thr, 64, columns=(4, 2, 3))
    st = plan.open()
    rows = st.consume(scan)
    counts, sums = st.finish()
    names = scan.dictionary(3)
    res = {names[g]: (int(counts[g]), int(counts[64 + g]), float(sums[g])) for g in range(len(names)) if counts[64 + g]}
    st.close()
    plan.close()
    flags = scan.decoded_on_gpu()
    scan.close()
    if want_gpu is not None:
        assert tuple(bool(x) for x in flags) == want_gpu, flags
    return rows, res


def same(a, b, rel=1e-12):
    assert a.keys() == b.keys()
    for k in b:
        assert a[k][:2] == b[k][:2], k
        assert a[k][2] == pytest.approx(b[k][2], rel=rel), k


@pytest.mark.parametrize("ragged,members", [(0, 1), (1, 1), (0, 3)])
def test_plain_fastq_gz_is_inflated_and_parsed_on_the_gpu(ctx, tmp_path, monkeypatch, ragged, members):
    n = 300_000
    path = tmp_path / "syn.fastq"
    subprocess.check_call([GEN, "fastq", str(n), str(path), "150", str(ragged)])
    gz, bgz = tmp_path / "plain.fastq.gz", tmp_path / "bgzf.fastq.gz"
    plain_gzip(path, gz, members=members)
    subprocess.check_call([BGZIP, str(path), str(bgz), "6"])
    monkeypatch.setenv("EXON_HIP_GZ_SLAB_MB", "16")  # several slabs: records and DEFLATE blocks straddle slab ends
    rows_g, gpu = hist(ctx, gz, True, want_gpu=(True, True))
    rows_b, twin = hist(ctx, bgz, True, want_gpu=(True, True))
    rows_p, plain = hist(ctx, path, True, want_gpu=(True, False))
    rows_h, host = hist(ctx, gz, False, want_gpu=(False, False))
    assert rows_g == rows_b == rows_p == rows_h == n
    assert np.array_equal(gpu, twin) and np.array_equal(gpu, plain) and np.array_equal(gpu, host)
    # EXON_HIP_GPU_GZIP=0: the host's zlib inflates, the GPU still parses
    monkeypatch.setenv("EXON_HIP_GPU_GZIP", "0")
    rows_z, viaz = hist(ctx, gz, True, want_gpu=(True, False))
    assert rows_z == n and np.array_equal(viaz, gpu)


def test_plain_vcf_gz_equals_the_oracle_and_its_twins(ctx, oracle, tmp_path, monkeypatch):
    n = 400_000
    path = tmp_path / "syn.vcf"
    subprocess.check_call([GEN, "vcf", str(n), str(path)])
    rows_o, want = k4_expected(oracle, path, "vcf", "AF")
    gz, bgz = tmp_path / "plain.vcf.gz", tmp_path / "bgzf.vcf.gz"
    plain_gzip(path, gz, level=6)
    subprocess.check_call([BGZIP, str(path), str(bgz), "6"])
    monkeypatch.setenv("EXON_HIP_GZ_SLAB_MB", "8")
    rows_g, gpu = k4(ctx, gz, True, want_gpu=(True, True))
    rows_b, twin = k4(ctx, bgz, True, want_gpu=(True, True))
    rows_h, host = k4(ctx, gz, False, want_gpu=(False, False))
    assert rows_g == rows_b == rows_h == rows_o == n
    same(gpu, want, rel=1e-9)
    same(gpu, twin)
    same(gpu, host)
    # levels 1 and 9 (other block sizes and match statistics), same answer
    for level in (1, 9):
        g2 = tmp_path / f"plain{level}.vcf.gz"
        plain_gzip(path, g2, level=level)
        rows_l, res = k4(ctx, g2, True, want_gpu=(True, True))
        assert rows_l == n
        same(res, gpu)


def test_reference_fixtures_plain_gzip(ctx, oracle, tmp_path):
    """the reference's own plain-gzip fixture (2 reads), and its VCF fixture re-compressed as ONE gzip member"""
    p = os.path.join(FX, "fastq", "test.fastq.gz")
    rows, gpu = hist(ctx, p, True, lmax=512, want_gpu=(True, True))
    off, data = decode.fastq_device_columns(decode.decode_fastq(os.path.join(FX, "fastq", "test.fastq")))
    h, _ = oracle.c5_qual_pos_hist(off, data, 512)
    assert rows == 2 and np.array_equal(gpu, h.reshape(-1))
    vcf = tmp_path / "index_plain.vcf.gz"
    with open(vcf, "wb") as f:
        f.write(gzip.compress(open(os.path.join(FX, "vcf", "index.vcf"), "rb").read(), 6))
    rows_o, want = k4_expected(oracle, os.path.join(FX, "vcf", "index.vcf"), "vcf", "MQ0F", thr=-1.0)
    rows_g, gpu = k4(ctx, vcf, True, want_gpu=(True, True), info_field="MQ0F", thr=-1.0)
    assert rows_g == rows_o == 621  # slt/vcf-select-tests.slt:47-55
    same(gpu, want, rel=1e-9)


def test_a_corrupt_plain_gzip_goes_to_the_host_reader(ctx, tmp_path):
    """a flipped byte in the DEFLATE data: the device decode does not prove, the state is rolled back, and the host reader reports
    what zlib finds (an error -- never a silently different answer)"""
    path = tmp_path / "syn.fastq"
    subprocess.check_call([GEN, "fastq", "50000", str(path), "100", "0"])
    gz = tmp_path / "bad.fastq.gz"
    plain_gzip(path, gz)
    raw = bytearray(open(gz, "rb").read())
    raw[len(raw) // 2] ^= 0x10
    open(gz, "wb").write(bytes(raw))
    with pytest.raises(exon_amd.ExonHipError):
        hist(ctx, gz, True)
    # a truncated file: same
    open(gz, "wb").write(bytes(raw[:len(raw) // 3]))
    with pytest.raises(exon_amd.ExonHipError):
        hist(ctx, gz, True)
