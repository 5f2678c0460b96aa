"""GPU-side SAM text parsing (exon_hip_sam_parser_*): alignment lines -> the BAM device layout -> K3 / K6, against the
ORACLE (oracle/decode.py decode_sam + the oracle's aggregates) and, as a second opinion, the native host SAM reader (same
columns as BAM, exon-sam/src/schema_builder.rs:371-402)."""
import os
import subprocess

import numpy as np
import pytest

import exon_amd
from oracle_expect import k3_expected, k6_expected

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")


def _k3(ctx, path, gpu_parse, fallback=False, compression=None, inflated=None, qmin=30):
    scan = exon_amd.Scan(str(path), "sam", gpu_parse=gpu_parse, compression=compression)
    refs = scan.dictionary(2)
    plan = ctx.plan_flag_mapq_group_count(1284, 0, qmin, len(refs), columns=(0, 1, 2))
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    assert scan.decoded_on_gpu()[0] == (bool(gpu_parse) and not fallback), "silent host fallback"
    if inflated is not None:
        assert scan.decoded_on_gpu()[1] == inflated
    st.close(); plan.close(); scan.close()
    return rows, np.array(counts)


def _k6(ctx, path, gpu_parse, ref, a, b):
    scan = exon_amd.Scan(str(path), "sam", gpu_parse=gpu_parse)
    plan = ctx.plan_overlap_count(scan.dictionary(2).index(ref), a, b)
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    st.close(); plan.close(); scan.close()
    return rows, int(counts[0])


def test_sam_reference_fixture(ctx, oracle):
    path = os.path.join(FX, "sam", "test.sam")
    g, h = _k3(ctx, path, True, qmin=0), _k3(ctx, path, False, qmin=0)
    rows_o, want = k3_expected(oracle, path, "sam", qmin=0)
    assert g[0] == rows_o == h[0] == 1 and np.array_equal(g[1], want) and np.array_equal(g[1], h[1]) and want.sum() == 1
    refs = exon_amd.Scan(path, "sam").dictionary(2)
    assert _k6(ctx, path, True, refs[0], 1, None) == k6_expected(path, "sam", refs[0], 1, None) == _k6(ctx, path, False, refs[0], 1, None)


def test_sam_file_to_gpu_pipeline_equals_the_oracle(ctx, oracle, tmp_path, monkeypatch):
    n = 120_000
    path = tmp_path / "syn.sam"
    subprocess.check_call([GEN, "sam", str(n), str(path), "100"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "1")
    g = _k3(ctx, path, True)
    rows_o, want = k3_expected(oracle, path, "sam")
    assert g[0] == rows_o == n and np.array_equal(g[1], want) and want.sum() > n // 4
    assert _k6(ctx, path, True, "chr7", 50_000_000, 100_000_000) == k6_expected(path, "sam", "chr7", 50_000_000, 100_000_000)


@pytest.mark.parametrize("slab_mb", ["1", "64"])
def test_sam_file_to_gpu_pipeline_equals_host_decode(ctx, tmp_path, monkeypatch, slab_mb):
    n = 300_000
    path = tmp_path / "syn.sam"
    subprocess.check_call([GEN, "sam", str(n), str(path), "100"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", slab_mb)
    g, h = _k3(ctx, path, True), _k3(ctx, path, False)
    assert g[0] == h[0] == n and np.array_equal(g[1], h[1]) and g[1].sum() > n // 4
    assert _k6(ctx, path, True, "chr7", 50_000_000, 100_000_000) == _k6(ctx, path, False, "chr7", 50_000_000, 100_000_000)


def test_sam_bgzf_is_inflated_and_parsed_on_the_gpu(ctx, tmp_path, monkeypatch):
    """bgzip-compressed SAM: the blocks are inflated by inflate.hip and the text never visits the host; a plain gzip
    member (not BGZF) is inflated on the GPU as well since round 6 (gzip_stream.hip) -- and by the host reader, still parsed on the
    device, with EXON_HIP_GPU_GZIP=0."""
    import gzip
    n = 100_000
    path = tmp_path / "syn.sam"
    subprocess.check_call([GEN, "sam", str(n), str(path), "100"])
    gz = tmp_path / "syn.sam.gz"
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "bgzip"), str(path), str(gz), "6"])
    h = _k3(ctx, path, False)
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "4")  # several slabs: records straddle inflated-slab boundaries
    g = _k3(ctx, gz, True, compression="gzip", inflated=True)
    assert g[0] == h[0] == n and np.array_equal(g[1], h[1])
    plain_gz = tmp_path / "member.sam.gz"
    with gzip.open(plain_gz, "wb", compresslevel=1) as f:
        f.write(open(path, "rb").read())
    g2 = _k3(ctx, plain_gz, True, compression="gzip", inflated=True)
    assert g2[0] == n and np.array_equal(g2[1], h[1])
    monkeypatch.setenv("EXON_HIP_GPU_GZIP", "0")
    g3 = _k3(ctx, plain_gz, True, compression="gzip", inflated=False)
    assert g3[0] == n and np.array_equal(g3[1], h[1])


def test_sam_lines_the_device_cannot_decide_fall_back(ctx, tmp_path):
    path = tmp_path / "odd.sam"
    with open(path, "w") as f:
        f.write("@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\n")
        for i in range(2000):
            flag = "99x" if i == 1500 else "99"  # atoi("99x") = 99 on the host; the device hands the file back
            f.write(f"r{i}\t{flag}\tchr1\t{i + 1}\t60\t50M\t*\t0\t0\t*\t*\n")
    g, h = _k3(ctx, path, True, fallback=True), _k3(ctx, path, False)
    assert g[0] == h[0] == 2000 and np.array_equal(g[1], h[1])
