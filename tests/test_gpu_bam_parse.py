"""GPU-side BAM path: BGZF inflate -> record splitting (parallel chain walk with proven guesses) -> field extraction -> K3,
against the ORACLE's decoder (oracle/decode.py: decode_bam restates the reference's builder) and, as a second opinion, the
product's native host decoder."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import exon_amd
from oracle_expect import bam_columns, k3_expected, k6_expected

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")


def bits(bm, n):
    return np.unpackbits(bm, bitorder="little")[:n].astype(bool)


def host_columns(path):
    s = exon_amd.Scan(path, "bam")
    batches = list(s)
    refs = s.dictionary(2)
    out = {k: [x for b in batches for x in b.field(i).to_pylist()] for i, k in enumerate(["flag", "mapq", "ref", "start", "end"])}
    s.close()
    return out, refs


def records_of(path):
    """(bytes of the record section, n_ref) of a BAM file."""
    raw = gzip.decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, o)[0]
        o += 4 + l_name + 4
    return raw[o:], n_ref


def check(res, host, refs):
    n = res["n_rows"]
    assert res["n_undecided"] == 0 and n == len(host["flag"])
    assert res["flag"].tolist() == host["flag"]
    mv, rv, pv = bits(res["mapq_valid"], n), bits(res["ref_valid"], n), bits(res["pos_valid"], n)
    assert [int(m) if v else None for m, v in zip(res["mapq"], mv)] == host["mapq"]
    assert [refs[r] if v else None for r, v in zip(res["ref_id"], rv)] == host["ref"]
    assert [int(x) if v else None for x, v in zip(res["start"], pv)] == host["start"]
    assert [int(x) if v else None for x, v in zip(res["end"], pv)] == host["end"]


def test_bam_parser_reference_fixture(ctx):
    path = os.path.join(FX, "bam", "test.bam")
    host, refs = host_columns(path)
    orefs, orc = bam_columns(path)                           # the oracle's decoder: the parity reference
    data, n_ref = records_of(path)
    p = exon_amd.BAMParser(ctx, n_ref, max_slab_bytes=1 << 20)
    res = p.parse_host(data)
    assert res["consumed_bytes"] == len(data) and res["n_rows"] == 61
    check(res, orc, orefs)
    check(res, host, refs)
    # first row of slt/bam-select-tests.slt:9-12: flag 83, chr1, 12203704..12217173, mapq NULL
    assert res["flag"][0] == 83 and res["start"][0] == 12203704 and res["end"][0] == 12217173 and not bits(res["mapq_valid"], 61)[0]
    p.close()


def test_bam_parser_many_segments_and_cut_off_record(ctx, tmp_path):
    ub = tmp_path / "syn.ubam"
    subprocess.check_call([GEN, "bam", "40000", str(ub), "100"])
    bam = tmp_path / "syn.bam"
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    host, refs = host_columns(str(bam))
    orefs, orc = bam_columns(str(bam))
    data, n_ref = records_of(str(bam))
    assert len(data) > 100 * 65536 // 2  # dozens of 64 KiB segments
    p = exon_amd.BAMParser(ctx, n_ref, max_slab_bytes=len(data) + 64)
    res = p.parse_host(data)
    assert res["consumed_bytes"] == len(data) and res["n_rows"] == 40000
    check(res, orc, orefs)
    check(res, host, refs)
    # a slab that stops in the middle of a record: whole records only, the cut-off one is left to the caller
    cut = len(data) - 37
    res = p.parse_host(data[:cut])
    assert res["n_undecided"] == 0 and res["n_rows"] == 39999 and 0 < res["consumed_bytes"] < cut
    bs = struct.unpack_from("<i", data, res["consumed_bytes"])[0]
    assert res["consumed_bytes"] + 4 + bs == len(data)
    # a slab cut in the middle of the 4-byte length field, one several segments before the end, and slabs whose last
    # 64 KiB segment holds only a few bytes of a cut-off record (or nothing at all)
    seg_edge = (len(data) // 65536) * 65536
    for cut in (res["consumed_bytes"] + 2, len(data) - 3 * 65536 - 11, seg_edge + 3, seg_edge + 20, seg_edge + 143, seg_edge, seg_edge - 1):
        r2 = p.parse_host(data[:cut])
        assert r2["n_undecided"] == 0 and r2["consumed_bytes"] <= cut
        assert r2["flag"].tolist() == host["flag"][:r2["n_rows"]]
    p.close()


def test_bam_parser_gives_up_on_malformed_input(ctx):
    path = os.path.join(FX, "bam", "test.bam")
    data, n_ref = records_of(path)
    p = exon_amd.BAMParser(ctx, n_ref, max_slab_bytes=1 << 20)
    bad = bytearray(data)
    bad[0:4] = struct.pack("<i", 7)  # block_size < 32
    assert p.parse_host(bytes(bad))["n_undecided"] > 0
    # a slab that does not start at a record boundary: the chain of segment 0 walks garbage
    big = bytearray(data * 8)  # ~190 KB: 3 segments
    res = p.parse_host(bytes(big[5:]))
    assert res["n_undecided"] > 0 or res["n_rows"] != 61 * 8
    p.close()


def _k3_through_scan(ctx, path, gpu_parse, fallback=False):
    scan = exon_amd.Scan(str(path), "bam", gpu_parse=gpu_parse)
    refs = scan.dictionary(2)
    plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, len(refs), columns=(0, 1, 2))
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    assert scan.decoded_on_gpu()[0] == (bool(gpu_parse) and not fallback), 'silent host fallback'
    st.close(); plan.close(); scan.close()
    return rows, np.array(counts)


@pytest.mark.parametrize("slab_mb", ["1", "64"])
def test_bam_file_to_gpu_pipeline_equals_host_decode(ctx, tmp_path, monkeypatch, slab_mb):
    n = 400_000
    ub = tmp_path / "syn.ubam"
    subprocess.check_call([GEN, "bam", str(n), str(ub), "100"])
    bam = tmp_path / "syn.bam"
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", slab_mb)  # "1": dozens of slabs, records carried across them
    rows_g, gpu = _k3_through_scan(ctx, bam, True)
    rows_h, host = _k3_through_scan(ctx, bam, False)
    assert rows_g == rows_h == n
    assert np.array_equal(gpu, host) and gpu.sum() > n // 4


def test_bam_reference_fixture_through_the_gpu_pipeline(ctx, oracle):
    path = os.path.join(FX, "bam", "test.bam")
    rows_g, gpu = _k3_through_scan(ctx, path, True)
    rows_o, want = k3_expected(oracle, path)                 # oracle decoder + oracle aggregate
    rows_h, host = _k3_through_scan(ctx, path, False)
    assert rows_g == rows_o == rows_h == 61 and np.array_equal(gpu, want) and np.array_equal(gpu, host)


def test_bam_file_to_gpu_pipeline_equals_the_oracle(ctx, oracle, tmp_path, monkeypatch):
    n = 120_000
    ub = tmp_path / "syn.ubam"
    subprocess.check_call([GEN, "bam", str(n), str(ub), "100"])
    bam = tmp_path / "syn.bam"
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "1")
    rows_g, gpu = _k3_through_scan(ctx, bam, True)
    rows_o, want = k3_expected(oracle, bam)
    assert rows_g == rows_o == n and np.array_equal(gpu, want) and gpu.sum() > n // 4
    assert _k6_through_scan(ctx, bam, True, "chr7", 50_000_000, 100_000_000) == k6_expected(bam, "bam", "chr7", 50_000_000, 100_000_000)


def _k6_through_scan(ctx, path, gpu_parse, region_ref, a, b):
    scan = exon_amd.Scan(str(path), "bam", gpu_parse=gpu_parse)
    rid = scan.dictionary(2).index(region_ref)
    plan = ctx.plan_overlap_count(rid, a, b)
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    st.close(); plan.close(); scan.close()
    return rows, int(counts[0])


def test_bam_region_overlap_on_the_gpu(ctx, tmp_path):
    """bam_region_filter('chr1:1-12209145', reference, start, end) on the reference fixture: 7 (slt/bam-indexed-select-
    tests.slt:16-19), file -> GPU inflate -> record split -> K6; and a synthetic BAM against the host decoder."""
    path = os.path.join(FX, "bam", "test.bam")
    assert _k6_through_scan(ctx, path, True, "chr1", 1, 12209145) == (61, 7)
    assert _k6_through_scan(ctx, path, False, "chr1", 1, 12209145) == (61, 7)
    ub = tmp_path / "syn.ubam"
    subprocess.check_call([GEN, "bam", "300000", str(ub), "100"])
    bam = tmp_path / "syn.bam"
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    g = _k6_through_scan(ctx, bam, True, "chr7", 50_000_000, 100_000_000)
    h = _k6_through_scan(ctx, bam, False, "chr7", 50_000_000, 100_000_000)
    assert g == h and g[0] == 300000 and g[1] > 1000


def _raw_bam_records(lengths, rng, n_ref=25):
    """Minimal well-formed BAM records with the given sequence lengths (1 CIGAR op, no aux)."""
    out = []
    for i, ln in enumerate(lengths):
        name = b"r%d\0" % i
        ref, pos = int(rng.integers(0, n_ref)), int(rng.integers(0, 1 << 28))
        body = struct.pack("<iiBBHHHiiii", ref, pos, len(name), int(rng.integers(0, 61)), 4680, 1, 0, ln, -1, -1, 0)
        body += name + struct.pack("<I", (ln << 4) | 0) + bytes((ln + 1) // 2) + bytes(ln)
        out.append(struct.pack("<i", len(body)) + body)
    return out


def test_bam_parser_long_reads_and_records_larger_than_a_segment(ctx):
    """Records of tens of kilobytes (long reads): segments with a single record start are still proven; a record longer
    than a 64 KiB segment makes the chain jump over whole segments, which the proof then ignores."""
    rng = np.random.default_rng(12)
    recs = _raw_bam_records([int(x) for x in rng.integers(20_000, 40_000, 200)], rng)
    data = b"".join(recs)
    p = exon_amd.BAMParser(ctx, 25, max_slab_bytes=len(data) + 200_000)
    res = p.parse_host(data)
    assert res["n_undecided"] == 0 and res["n_rows"] == 200 and res["consumed_bytes"] == len(data)
    lens = [struct.unpack_from("<i", r, 20)[0] for r in recs]
    pos = [struct.unpack_from("<i", r, 8)[0] for r in recs]
    assert res["start"].tolist() == [x + 1 for x in pos] and (res["end"] - res["start"] + 1).tolist() == lens
    lens = [1000, 90_000, 1000, 300_000, 5, 70_000, 1000]  # 135 KB / 450 KB records: whole segments inside one record
    big = b"".join(_raw_bam_records(lens, rng))
    p2 = exon_amd.BAMParser(ctx, 25, max_slab_bytes=len(big) + 200_000)
    res = p2.parse_host(big)
    assert res["n_undecided"] == 0 and res["n_rows"] == len(lens) and res["consumed_bytes"] == len(big)
    assert (res["end"] - res["start"] + 1).tolist() == lens
    res = p2.parse_host(big[:-100])  # the last (70 KB) record is cut off
    assert res["n_undecided"] == 0 and res["n_rows"] == len(lens) - 1
    p.close()
    p2.close()


def test_empty_inputs_through_the_gpu_pipelines(ctx, tmp_path):
    """Header-only BAM / VCF.gz and an empty FASTQ.gz: zero rows, no error, no fallback needed."""
    ub = tmp_path / "e.ubam"
    subprocess.check_call([GEN, "bam", "0", str(ub), "100"])
    bam = tmp_path / "e.bam"
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    rows, counts = _k3_through_scan(ctx, bam, True)
    assert rows == 0 and counts.sum() == 0
    vcf = tmp_path / "e.vcf"
    subprocess.check_call([GEN, "vcf", "0", str(vcf)])
    gz = tmp_path / "e.vcf.gz"
    subprocess.check_call([BGZIP, str(vcf), str(gz), "6"])
    scan = exon_amd.Scan(str(gz), "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 8, columns=(4, 2, 3))
    st = plan.open()
    assert st.consume(scan) == 0
    c, s_ = st.finish()
    assert sum(c) == 0
    st.close(); plan.close(); scan.close()
    fq = tmp_path / "e.fastq"
    fq.write_bytes(b"")
    fqgz = tmp_path / "e.fastq.gz"
    subprocess.check_call([BGZIP, str(fq), str(fqgz), "6"])
    scan = exon_amd.Scan(str(fqgz), "fastq", gpu_parse=True)
    plan = ctx.plan_qual_pos_hist(64, columns=(3,))
    st = plan.open()
    assert st.consume(scan) == 0
    st.close(); plan.close(); scan.close()
