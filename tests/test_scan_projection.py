"""The reference's projected schema out of the scans (exon_hip_scan_options.projection): VCF id / ref / alt
(exon-vcf/src/array_builder/lazy_array_builder.rs:169-205) and BAM name / cigar / sequence / quality_scores
(exon-bam/src/array_builder.rs:105-201), from the host readers (CPU tests) and from the GPU decode pipeline (-m gpu), column by column
against oracle/decode.py; the reference's pinned first BAM row (slt/bam-select-tests.slt:9-35) and the shape of its benchmark query
(exon-benchmarks/src/main.rs:143-157: SELECT chrom, pos, id ... WHERE vcf_region_filter)."""
import os
import subprocess

import numpy as np
import pytest

import exon_amd
from oracle import decode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")


def table(scan):
    import pyarrow as pa
    batches = list(scan)
    cols = {}
    for b in batches:
        for i in range(b.type.num_fields):
            cols.setdefault(b.type.field(i).name, []).extend(b.field(i).to_pylist())
    return cols


def check_vcf(cols, v):
    assert cols["chrom"] == v["chrom"] and cols["pos"] == v["pos"]
    assert cols["id"] == v["id"]
    assert cols["ref"] == v["ref"]
    assert cols["alt"] == v["alt"]


def check_bam(cols, refs, recs):
    assert cols["name"] == [r["name"] if r["name"] != "*" else None for r in recs]
    assert cols["cigar"] == [r["cigar"] for r in recs]
    assert cols["sequence"] == [r["sequence"] for r in recs]
    assert cols["quality_score"] == [[q - 256 if q > 127 else q for q in r["quality_score"]] for r in recs]
    assert cols["flag"] == [r["flag"] for r in recs]


def test_host_vcf_id_ref_alt_on_the_fixture():
    p = os.path.join(FX, "vcf", "index.vcf")
    v = decode.decode_vcf(p)
    for path in (p, p + ".gz"):
        s = exon_amd.Scan(path, "vcf", batch_size=100, project=("id", "ref", "alt"))
        names = [s.schema().field(i).name for i in range(s.schema().num_fields)]
        assert names[-3:] == ["id", "ref", "alt"]
        check_vcf(table(s), v)
        s.close()
    # a subset keeps the order of the bits; none = the default columns only
    s = exon_amd.Scan(p, "vcf", project=("ref",))
    assert [s.schema().field(i).name for i in range(s.schema().num_fields)] == ["chrom", "pos", "qual", "filter", "ref"]
    s.close()


def test_host_vcf_ids_lists_and_missing(tmp_path):
    p = tmp_path / "ids.vcf"
    head = "##fileformat=VCFv4.2\n##contig=<ID=1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    rows = ["1\t10\t.\tA\tC\t.\t.\t.", "1\t11\trs1\tAC\tA,ACC\t5\tPASS\t.", "1\t12\trs2;rs3;x\tG\t.\t.\t.\t.", "1\t13\trs9\tT\t<DEL>\t.\t.\t.\r"]
    p.write_text(head + "\n".join(rows) + "\n")
    c = table(exon_amd.Scan(str(p), "vcf", project=("id", "ref", "alt")))
    assert c["id"] == [None, ["rs1"], ["rs2", "rs3", "x"], ["rs9"]]
    assert c["ref"] == ["A", "AC", "G", "T"]
    assert c["alt"] == [[], [], None, []]  # the reference's alt lists have no items (lazy_array_builder.rs:191-205)
    v = decode.decode_vcf(str(p))
    check_vcf(c, v)


def test_host_bam_text_columns_on_the_fixture():
    p = os.path.join(FX, "bam", "test.bam")
    refs, recs = decode.decode_bam(p)
    s = exon_amd.Scan(p, "bam", batch_size=16, project=("name", "cigar", "sequence", "quality_score"))
    c = table(s)
    check_bam(c, refs, recs)
    # slt/bam-select-tests.slt:9-35: the first row, and quality_scores[1..5]
    assert (c["name"][0], c["flag"][0], c["start"][0], c["end"][0], c["mapping_quality"][0], c["cigar"][0]) == ("READ_ID", 83, 12203704, 12217173, None, "55M13394N21M")
    assert [q[0] for q in c["quality_score"][:5]] == [23, 20, 37, 34, 31] and [len(q) for q in c["quality_score"][:5]] == [76] * 5
    s.close()


def test_projection_is_refused_where_it_is_not_built():
    import ctypes as C
    from exon_amd import _lib as L
    lib = L.load()
    opt = L.ScanOptions(L.FORMATS["fastq"], 0, 0, None, None, 0, 0, 1)
    h = C.c_void_p()
    assert lib.exon_hip_scan_open(os.path.join(FX, "fastq", "test.fastq").encode(), C.byref(opt), C.byref(h)) == -4
    opt = L.ScanOptions(L.FORMATS["vcf"], 0, 0, None, None, 0, 0, 64)
    assert lib.exon_hip_scan_open(os.path.join(FX, "vcf", "index.vcf").encode(), C.byref(opt), C.byref(h)) == -1


# ---- the GPU decode pipeline ----------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["index.vcf", "index.vcf.gz"])
def test_gpu_vcf_id_ref_alt_on_the_fixture(ctx, name):
    p = os.path.join(FX, "vcf", name)
    v = decode.decode_vcf(p)
    s = exon_amd.Scan(p, "vcf", batch_size=100, gpu_parse=True, project=("id", "ref", "alt")).bind_ctx(ctx)
    c = table(s)
    assert s.decoded_on_gpu()[0]
    check_vcf(c, v)
    assert len(c["chrom"]) == 621
    s.close()


@pytest.mark.gpu
def test_gpu_vcf_text_columns_equal_the_host_readers_on_synthetic_rows(ctx, tmp_path, monkeypatch):
    """1 M synthetic rows whose ID fields are rewritten (missing / one / several ids) and REF / ALT of varying length, BGZF: the
    device columns equal the host reader's and the oracle's, across slabs; then the reference's benchmark query shape:
    chrom, pos, id of the rows that hit a region (the mask is applied on the device, the strings of the kept rows are gathered)."""
    n = 1_000_000
    base = tmp_path / "syn.vcf"
    subprocess.check_call([GEN, "vcf", str(n), str(base)])
    rng = np.random.default_rng(5)
    out = tmp_path / "ids.vcf"
    bases = ["A", "C", "G", "T", "AC", "GTT", "ACGTACGT"]
    with open(base) as f, open(out, "w") as g:
        i = 0
        for line in f:
            if line.startswith("#"):
                g.write(line)
                continue
            c = line.split("\t")
            k = i % 7
            c[2] = "." if k < 3 else f"rs{i}" if k < 6 else f"rs{i};ss{i * 3};x"
            c[3] = bases[i % len(bases)]
            c[4] = "." if i % 11 == 0 else bases[(i * 5) % len(bases)] + ("," + bases[i % 3] if i % 4 == 0 else "")
            g.write("\t".join(c))
            i += 1
    gz = str(out) + ".gz"
    subprocess.check_call([BGZIP, str(out), gz, "6"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "8")
    s = exon_amd.Scan(gz, "vcf", gpu_parse=True, project=("id", "ref", "alt")).bind_ctx(ctx)
    dev = table(s)
    assert s.decoded_on_gpu() == (True, True)
    s.close()
    host = table(exon_amd.Scan(gz, "vcf", project=("id", "ref", "alt")))
    assert len(dev["id"]) == n
    for k in ("chrom", "pos", "id", "ref", "alt"):
        assert dev[k] == host[k], k
    v = decode.decode_vcf(str(out))
    assert dev["id"] == v["id"] and dev["ref"] == v["ref"] and dev["alt"] == v["alt"]
    # SELECT chrom, pos, id FROM t WHERE vcf_region_filter('<chrom>:<lo>-<hi>', chrom, pos)
    c0 = v["chrom"][n // 2]
    ps = sorted(p for c, p in zip(v["chrom"], v["pos"]) if c == c0 and p is not None)
    lo, hi = ps[len(ps) // 4], ps[len(ps) // 4 + 5000]
    s = exon_amd.Scan(gz, "vcf", gpu_parse=True, region=f"{c0}:{lo}-{hi}", project=("id",)).bind_ctx(ctx)
    hit = table(s)
    s.close()
    want = [(c, p, d) for c, p, d in zip(v["chrom"], v["pos"], v["id"]) if c == c0 and p is not None and lo <= p <= hi]
    assert list(zip(hit["chrom"], hit["pos"], hit["id"])) == want and len(want) > 100


@pytest.mark.gpu
def test_gpu_bam_text_columns(ctx, tmp_path, monkeypatch):
    p = os.path.join(FX, "bam", "test.bam")
    refs, recs = decode.decode_bam(p)
    s = exon_amd.Scan(p, "bam", batch_size=16, gpu_parse=True, project=("name", "cigar", "sequence", "quality_score")).bind_ctx(ctx)
    c = table(s)
    assert s.decoded_on_gpu()[0]
    s.close()
    check_bam(c, refs, recs)
    assert (c["name"][0], c["flag"][0], c["start"][0], c["end"][0], c["mapping_quality"][0], c["cigar"][0]) == ("READ_ID", 83, 12203704, 12217173, None, "55M13394N21M")
    assert [q[0] for q in c["quality_score"][:5]] == [23, 20, 37, 34, 31] and [len(q) for q in c["quality_score"][:5]] == [76] * 5
    # 200 k synthetic reads over several slabs: device = host reader = oracle
    ub, bam = tmp_path / "syn.ubam", tmp_path / "syn.bam"
    subprocess.check_call([GEN, "bam", "200000", str(ub), "100"])
    subprocess.check_call([BGZIP, str(ub), str(bam), "6"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "8")
    proj = ("name", "cigar", "sequence", "quality_score")
    s = exon_amd.Scan(str(bam), "bam", gpu_parse=True, project=proj).bind_ctx(ctx)
    dev = table(s)
    s.close()
    host = table(exon_amd.Scan(str(bam), "bam", project=proj))
    for k in ("flag", "start", "end") + proj:
        assert dev[k] == host[k], k
    refs, recs = decode.decode_bam(str(bam))
    check_bam(dev, refs, recs)
    # a subset
    s = exon_amd.Scan(str(bam), "bam", gpu_parse=True, project=("cigar",)).bind_ctx(ctx)
    one = table(s)
    s.close()
    assert one["cigar"] == host["cigar"] and "name" not in one
