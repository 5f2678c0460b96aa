"""Pushed-down region filters (vcf_region_filter / bam_region_filter, INDEXED_VCF / INDEXED_BAM) on the GPU decode path:
the host plans the tabix / BAI chunks, only their BGZF blocks cross PCIe, the device inflates, parses, applies the per-record
interval hit (k_region_mask) and aggregates.  Pinned on the reference's slt values (191 / 382 / 11 / 0 for VCF, 7 / 14 for
BAM: slt/vcf-indexed-tests.slt:22-59, slt/bam-indexed-select-tests.slt:11-50) and, on synthetic files whose chunks start
and end inside BGZF blocks, compared with the host decoders and with a brute-force count."""
import os
import struct
import subprocess

import numpy as np
import pytest

import exon_amd
from bgzf_index_writer import sorted_bam, write_bai, write_tabix

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")


def vcf_region_rows(ctx, path, region, use_index, gpu_parse, fmt="vcf"):
    """(rows the scan emitted, COUNT(*) of a plan that keeps every emitted row, decoded on the GPU?, index chunks)"""
    scan = exon_amd.Scan(path, fmt, region=region, use_index=use_index, gpu_parse=gpu_parse)
    contigs = scan.dictionary(0)
    name = region.split(":")[0]
    # chrom = <region's contig> AND pos >= 1: every row the region filter lets through satisfies it
    plan = ctx.plan_region_count(contigs.index(name) if name in contigs else 0, 1, None, columns=(0, 1))
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    dec = scan.decoded_on_gpu()
    chunks = scan.index_chunks()
    st.close(); plan.close(); scan.close()
    return rows, int(counts[0]), dec[0], chunks


@pytest.mark.parametrize("path,region,want", [
    ("vcf/index.vcf.gz", "1", 191),                    # slt/vcf-indexed-tests.slt:31-35 (one of the two partitions)
    ("vcf-partition/sample=1/index1.vcf.gz", "1", 191),
    ("vcf-partition/sample=2/index2.vcf.gz", "1", 191),  # 382 over both: :22-28
    ("vcf/index.vcf.gz", "a", 0),                      # :38-43
    ("biobear-vcf/vcf_file.vcf.gz", "1", 11),          # :46-50
    ("biobear-vcf/vcf_file.vcf.gz", "1000", 0),        # :52-56
])
def test_reference_vcf_region_pins_through_the_gpu_indexed_path(ctx, path, region, want):
    p = os.path.join(FX, path)
    if not os.path.exists(p):
        pytest.skip(f"fixture {path} not present")
    rows, cnt, on_gpu, chunks = vcf_region_rows(ctx, p, region, use_index=True, gpu_parse=True)
    assert rows == want and on_gpu and chunks >= (1 if want else 0)
    if want:
        assert cnt == want
    # and the same through the host decoder, and without the index (whole file parsed on the GPU, rows masked)
    assert vcf_region_rows(ctx, p, region, use_index=True, gpu_parse=False)[0] == want
    rows_u, _, on_gpu_u, chunks_u = vcf_region_rows(ctx, p, region, use_index=False, gpu_parse=True)
    assert rows_u == want and on_gpu_u and chunks_u == -1


def bam_region_rows(ctx, path, region, use_index, gpu_parse):
    scan = exon_amd.Scan(path, "bam", region=region, use_index=use_index, gpu_parse=gpu_parse)
    refs = scan.dictionary(2)
    name = region.split(":")[0]
    plan = ctx.plan_overlap_count(refs.index(name) if name in refs else 0, 1, None)  # keeps every row the filter emits
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    dec = scan.decoded_on_gpu()
    chunks = scan.index_chunks()
    st.close(); plan.close(); scan.close()
    return rows, int(counts[0]), dec[0], chunks


def test_reference_bam_region_pins_through_the_gpu_indexed_path(ctx):
    one = os.path.join(FX, "bam", "test.bam")
    rows, cnt, on_gpu, chunks = bam_region_rows(ctx, one, "chr1:1-12209145", True, True)
    assert (rows, cnt, on_gpu) == (7, 7, True) and chunks >= 1       # slt/bam-indexed-select-tests.slt:16-19
    assert bam_region_rows(ctx, one, "chr1:1-12209145", True, False)[:2] == (7, 7)
    assert bam_region_rows(ctx, one, "chr1:1-12209145", False, True)[:3] == (7, 7, True)
    total = 0
    for f in ("test.bam", "test2.bam"):                               # two files: 14 (:29-33)
        total += bam_region_rows(ctx, os.path.join(FX, "bam-multifile", f), "chr1:1-12209145", True, True)[0]
    assert total == 14
    assert bam_region_rows(ctx, one, "chrZ:1-100", True, True)[0] == 0  # a reference the header does not know


def k4_region(ctx, path, region, use_index, gpu_parse):
    scan = exon_amd.Scan(path, "vcf", info_field="AF", region=region, use_index=use_index, gpu_parse=gpu_parse)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    st = plan.open()
    rows = st.consume(scan)
    counts, sums = st.finish()
    names = scan.dictionary(3)
    res = {names[g]: (int(counts[g]), int(counts[64 + g]), float(sums[g])) for g in range(len(names)) if counts[64 + g]}
    dec = scan.decoded_on_gpu()[0]
    st.close(); plan.close(); scan.close()
    return rows, res, dec


@pytest.mark.parametrize("slab_mb", ["4", "64"])
def test_synthetic_indexed_vcf_chunks_inside_blocks(ctx, tmp_path, monkeypatch, slab_mb):
    """1.2 M records in ~1000 BGZF blocks with a tabix index written by the test: regions that start and end inside blocks,
    at the first and the last record, beyond the data; the mask is ANDed with the validity of info.AF (the plan's first
    operand).  GPU indexed == host indexed == GPU unindexed (.vcf.gz and plain text) == brute force over the generator."""
    n = 1_200_000
    path = tmp_path / "syn.vcf"
    subprocess.check_call([GEN, "vcf", str(n), str(path)])
    gz = tmp_path / "syn.vcf.gz"
    subprocess.check_call([BGZIP, str(path), str(gz), "6"])
    assert write_tabix(gz) == n
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", slab_mb)
    for region, want_rows in [("1:500000-600000", 100_001), ("1:1-10", 10), (f"1:{n - 4}-{n + 100}", 5), ("1", n),
                              ("1:777777", n - 777_776), ("1:99999999-100000000", 0), ("zz:1-5", 0)]:
        g = k4_region(ctx, gz, region, True, True)
        h = k4_region(ctx, gz, region, True, False)
        u = k4_region(ctx, gz, region, False, True)
        t = k4_region(ctx, path, region, False, True)
        assert g[0] == h[0] == u[0] == t[0] == want_rows, region
        assert g[2] and u[2] and t[2] and not h[2], region
        for other in (h[1], u[1], t[1]):
            assert g[1].keys() == other.keys()
            for k in other:
                assert g[1][k][:2] == other[k][:2]
                assert g[1][k][2] == pytest.approx(other[k][2], rel=1e-12)


def test_synthetic_indexed_bam_chunks_inside_blocks(ctx, tmp_path, monkeypatch):
    rng = np.random.default_rng(21)
    n = 400_000
    ub = tmp_path / "s.ubam"
    rows = sorted_bam(ub, n, rng)
    bam = tmp_path / "s.bam"
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    assert write_bai(bam) == n - n // 50
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "4")
    arr = np.array(rows, np.int64)
    for region, (rid, a, b) in [("chr2:10000000-20000000", (1, 10_000_000, 20_000_000)), ("chr1:1-5000", (0, 1, 5000)),
                                ("chr4:48999000-60000000", (3, 48_999_000, 60_000_000)), ("chr3", (2, 1, 2**62))]:
        want = int(((arr[:, 0] == rid) & (arr[:, 1] <= b) & (arr[:, 2] >= a)).sum())
        g = bam_region_rows(ctx, bam, region, True, True)
        h = bam_region_rows(ctx, bam, region, True, False)
        u = bam_region_rows(ctx, bam, region, False, True)
        assert g[0] == h[0] == u[0] == want and g[1] == h[1] == u[1] == want, region
        assert g[2] and u[2] and not h[2] and g[3] >= 1, region


def test_region_mask_on_sam_and_bcf_full_scans(ctx):
    """No index for these formats in the reference's GPU-relevant paths: the whole file is parsed on the device and masked."""
    bcf = os.path.join(FX, "bcf", "index.bcf")
    rows, cnt, on_gpu, _ = vcf_region_rows(ctx, bcf, "1", use_index=False, gpu_parse=True, fmt="bcf")
    assert (rows, cnt, on_gpu) == (191, 191, True)        # exon_context_ext.rs:1053-1090: 191 records of region '1'
    sam = os.path.join(FX, "sam", "test.sam")  # one alignment: ref1, POS 1, 10M
    for region, want in (("ref1:1-5", 1), ("ref1:11-20", 0), ("ref1", 1), ("other", 0)):
        got = []
        for gpu in (True, False):
            scan = exon_amd.Scan(sam, "sam", region=region, gpu_parse=gpu)
            plan = ctx.plan_overlap_count(0, 1, None)
            st = plan.open()
            rows = st.consume(scan)
            counts, _ = st.finish()
            assert scan.decoded_on_gpu()[0] == gpu
            st.close(); plan.close(); scan.close()
            got.append((rows, int(counts[0])))
        assert got[0] == got[1] == (want, want), region


@pytest.mark.parametrize("rel", ["vcf/index.vcf.gz", "biobear-vcf/vcf_file.vcf.gz"])
def test_every_contig_count_equals_the_htslib_written_index(ctx, rel):
    """The tabix indexes beside the reference's fixtures carry htslib's own per-contig record counts (pseudo-bin 37450): 191 / 219 /
    211 and 11 / 1 / 1 / 2.  The slt files pin one contig of each; through the GPU indexed path and through the unindexed GPU
    parse + row mask EVERY contig must come out with htslib's count."""
    from index_meta import tabix_counts
    p = os.path.join(FX, rel)
    want = tabix_counts(p + ".tbi")
    assert len(want) >= 3
    for contig, n in want.items():
        rows, cnt, on_gpu, chunks = vcf_region_rows(ctx, p, contig, use_index=True, gpu_parse=True)
        assert (rows, cnt, on_gpu) == (n, n, True) and chunks >= 1, contig
        assert vcf_region_rows(ctx, p, contig, use_index=False, gpu_parse=True)[0] == n, contig
