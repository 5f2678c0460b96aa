"""GPU-side BCF2 path: BGZF inflate -> record splitting (proven chain walk) -> typed-value decode -> fused kernels,
against the ORACLE (oracle/decode.py decode_bcf + the oracle's aggregates; pinned on the reference's 621 / 191), the native
host BCF decoder and the VCF twin of the same rows."""
import os
import subprocess

import numpy as np
import pytest

import exon_amd
from oracle_expect import k4_expected, region_count_expected

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")


def _k4(ctx, path, fmt, gpu_parse, info_field="AF", fallback=False, thr=0.01):
    scan = exon_amd.Scan(str(path), fmt, info_field=info_field, gpu_parse=gpu_parse)
    plan = ctx.plan_cmp_avg_by_group(">", thr, 64, columns=(4, 2, 3))
    st = plan.open()
    rows = st.consume(scan)
    counts, sums = st.finish()
    names = scan.dictionary(3)
    res = {names[g]: (int(counts[g]), int(counts[64 + g]), float(sums[g])) for g in range(len(names)) if counts[64 + g]}
    assert scan.decoded_on_gpu()[0] == (bool(gpu_parse) and not fallback), 'silent host fallback'
    st.close(); plan.close(); scan.close()
    return rows, res


def _region_count(ctx, path, gpu_parse, chrom):
    scan = exon_amd.Scan(str(path), "bcf", gpu_parse=gpu_parse)
    plan = ctx.plan_region_count(scan.dictionary(0).index(chrom), 1, None, columns=(0, 1))
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    st.close(); plan.close(); scan.close()
    return rows, int(counts[0])


def test_bcf_reference_fixture_through_the_gpu(ctx, oracle):
    """index.bcf: 621 records, 191 on chromosome "1" (exon_context_ext.rs:1053-1090), decoded on the GPU, by the oracle and
    on the host."""
    path = os.path.join(FX, "bcf", "index.bcf")
    assert _region_count(ctx, path, True, "1") == region_count_expected(path, "bcf", "1") == (621, 191)
    assert _region_count(ctx, path, False, "1") == (621, 191)
    for field, thr in (("MQ0F", -1.0), ("DP", 3.0)):        # oracle decoder + oracle aggregate over the same file
        rows_o, want = k4_expected(oracle, path, "bcf", field, thr=thr)
        g = _k4(ctx, path, "bcf", True, field, thr=thr)
        assert g[0] == rows_o == 621 and g[1].keys() == want.keys()
        for k in want:
            assert g[1][k][:2] == want[k][:2] and g[1][k][2] == pytest.approx(want[k][2], rel=1e-9)
    # a typed INFO field + GROUP BY filter through both decoders
    for field in ("MQ0F", "DP"):
        g = _k4(ctx, path, "bcf", True, field)
        h = _k4(ctx, path, "bcf", False, field)
        assert g[0] == h[0] == 621 and g[1].keys() == h[1].keys()
        for k in h[1]:
            assert g[1][k][:2] == h[1][k][:2] and g[1][k][2] == pytest.approx(h[1][k][2], rel=1e-12)


@pytest.mark.parametrize("slab_mb", ["1", "64"])
def test_bcf_file_to_gpu_pipeline_equals_host_and_vcf_twin(ctx, tmp_path, monkeypatch, slab_mb):
    n = 1_200_000
    ub, bcf, vcf = tmp_path / "syn.ubcf", tmp_path / "syn.bcf", tmp_path / "syn.vcf"
    subprocess.check_call([GEN, "bcf", str(n), str(ub)])
    subprocess.check_call([BGZIP, str(ub), str(bcf), "6"])
    subprocess.check_call([GEN, "vcf", str(n), str(vcf)])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", slab_mb)  # "1": dozens of slabs, records carried across them
    rows_g, gpu = _k4(ctx, bcf, "bcf", True)
    rows_h, host = _k4(ctx, bcf, "bcf", False)
    rows_v, twin = _k4(ctx, vcf, "vcf", True)
    assert rows_g == rows_h == rows_v == n
    for other in (host, twin):
        assert gpu.keys() == other.keys()
        for k in other:
            assert gpu[k][:2] == other[k][:2]
            assert gpu[k][2] == pytest.approx(other[k][2], rel=1e-12)
    assert set(gpu) == {"PASS", "", "q10", "q10;s50", "s50"}


def test_bcf_file_to_gpu_pipeline_equals_the_oracle(ctx, oracle, tmp_path, monkeypatch):
    n = 150_000
    ub, bcf = tmp_path / "syn.ubcf", tmp_path / "syn.bcf"
    subprocess.check_call([GEN, "bcf", str(n), str(ub)])
    subprocess.check_call([BGZIP, str(ub), str(bcf), "6"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "1")
    rows_g, gpu = _k4(ctx, bcf, "bcf", True)
    rows_o, want = k4_expected(oracle, bcf, "bcf", "AF")
    assert rows_g == rows_o == n and gpu.keys() == want.keys()
    for k in want:
        assert gpu[k][:2] == want[k][:2] and gpu[k][2] == pytest.approx(want[k][2], rel=1e-9)
