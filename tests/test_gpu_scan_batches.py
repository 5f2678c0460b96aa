"""Batches from the GPU decode pipeline (exon_hip_scan_bind_ctx + exon_hip_scan_next): the surface of <Fmt>Scan::execute
(exon-core/src/datasources/vcf/scanner.rs:142-162, bam/scanner.rs:138-158) for queries that do not end in a fused kernel.  Every
batch column must equal what the host reader builds from the same file (which tests/test_scan_decoders.py pins on the oracle),
batch boundaries included."""
import os
import subprocess

import numpy as np
import pytest

import exon_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")


def columns(batches):
    n = batches[0].type.num_fields if batches else 0
    return [[x for b in batches for x in b.field(c).to_pylist()] for c in range(n)], [len(b) for b in batches]


def host_and_gpu(ctx, path, fmt, **kw):
    h = exon_amd.Scan(path, fmt, **kw)
    want, want_sizes = columns(list(h))
    h.close()
    g = exon_amd.Scan(path, fmt, gpu_parse=True, **kw).bind_ctx(ctx)
    got, got_sizes = columns(list(g))
    flags = g.decoded_on_gpu()
    rows = g.rows()
    g.close()
    return want, want_sizes, got, got_sizes, flags, rows


def same(want, got):
    assert len(want) == len(got)
    for c, (a, b) in enumerate(zip(want, got)):
        if a and isinstance(next((x for x in a if x is not None), None), float):
            a2 = np.array([np.nan if x is None else x for x in a], np.float32)
            b2 = np.array([np.nan if x is None else x for x in b], np.float32)
            assert np.array_equal(a2.view(np.uint32), b2.view(np.uint32)), c
            assert [x is None for x in a] == [x is None for x in b], c
        else:
            assert a == b, c


@pytest.mark.parametrize("kind", ["vcf", "vcf.gz", "bcf", "bam", "sam"])
def test_batches_from_the_gpu_pipeline_equal_the_host_readers(ctx, tmp_path, kind):
    n = 60_000
    if kind in ("vcf", "vcf.gz"):
        path = str(tmp_path / "t.vcf")
        subprocess.check_call([GEN, "vcf", str(n), path], stdout=subprocess.DEVNULL)
        if kind == "vcf.gz":
            subprocess.check_call([BGZIP, path, path + ".gz", "6"], stdout=subprocess.DEVNULL)
            path += ".gz"
        fmt, kw = "vcf", {"info_field": "AF"}
    elif kind == "bcf":
        ub, path = str(tmp_path / "t.ubcf"), str(tmp_path / "t.bcf")
        subprocess.check_call([GEN, "bcf", str(n), ub], stdout=subprocess.DEVNULL)
        subprocess.check_call([BGZIP, ub, path, "6"], stdout=subprocess.DEVNULL)
        fmt, kw = "bcf", {"info_field": "AF"}
    elif kind == "bam":
        ub, path = str(tmp_path / "t.ubam"), str(tmp_path / "t.bam")
        subprocess.check_call([GEN, "bam", str(n), ub], stdout=subprocess.DEVNULL)
        subprocess.check_call([BGZIP, ub, path, "6"], stdout=subprocess.DEVNULL)
        fmt, kw = "bam", {}
    else:
        path = str(tmp_path / "t.sam")
        subprocess.check_call([GEN, "sam", str(n), path], stdout=subprocess.DEVNULL)
        fmt, kw = "sam", {}
    want, want_sizes, got, got_sizes, flags, rows = host_and_gpu(ctx, path, fmt, batch_size=5000, **kw)
    assert rows == n and sum(got_sizes) == n and flags[0], "the batches did not come from the GPU pipeline"
    assert flags[1] == (kind in ("vcf.gz", "bcf", "bam"))
    same(want, got)
    assert all(k <= 5000 for k in got_sizes)


def test_reference_fixture_batches_and_region_filter(ctx):
    """index.vcf.gz of the reference: 621 rows, 191 on chromosome '1' (exon_context_ext.rs:1053-1090) -- whole file and with the
    pushed-down region, typed INFO keys of three kinds, through the GPU pipeline = the host reader."""
    path = os.path.join(FX, "vcf", "index.vcf.gz")
    for kw in ({}, {"region": "1"}, {"region": "1:10000-100000"}):
        want, want_sizes, got, sizes, flags, rows = host_and_gpu(ctx, path, "vcf", info_field="DP,MQ0F,INDEL", **kw)
        assert flags[0] and rows == sum(want_sizes) == sum(sizes)
        if rows:
            same(want, got)
    assert host_and_gpu(ctx, path, "vcf", region="1")[5] == 191
    path = os.path.join(FX, "bam", "test.bam")
    want, _, got, sizes, flags, rows = host_and_gpu(ctx, path, "bam")
    assert flags[0] and rows == sum(sizes)
    same(want, got)


def test_rows_the_device_cannot_decide_come_from_the_host_reader_in_order(ctx, tmp_path):
    """A contig the header does not declare, deep inside the file: the GPU pipeline hands over, the host reader continues behind the
    rows already emitted -- same batches' content as the host reader alone, nothing twice, nothing lost."""
    path = str(tmp_path / "t.vcf")
    subprocess.check_call([GEN, "vcf", "40000", path], stdout=subprocess.DEVNULL)
    lines = open(path).read().split("\n")
    body = [i for i, ln in enumerate(lines) if ln and not ln.startswith("#")]
    k = body[31_000]
    lines[k] = "chrUn_x" + lines[k][lines[k].index("\t"):]
    open(path, "w").write("\n".join(lines))
    want, _, got, sizes, flags, rows = host_and_gpu(ctx, path, "vcf", info_field="AF", batch_size=3000)
    assert rows == 40000 and not flags[0]
    same(want, got)


def test_scan_closed_with_batches_outstanding(ctx, tmp_path):
    path = str(tmp_path / "t.vcf")
    subprocess.check_call([GEN, "vcf", "300000", path], stdout=subprocess.DEVNULL)
    g = exon_amd.Scan(path, "vcf", gpu_parse=True, batch_size=1000).bind_ctx(ctx)
    it = iter(g)
    assert len(next(it)) == 1000
    g.close()                      # the producer is blocked on a full queue: close() must wake and join it
    with pytest.raises(exon_amd.ExonHipError):
        exon_amd.Scan(path, "fasta", gpu_parse=True).bind_ctx(ctx)


# ---- FASTQ batches from the GPU pipeline (round 6): the four Utf8 columns built on the device (text_columns.hip) --------------------
def _fastq_cols(scan):
    out = {}
    for b in scan:
        for i in range(b.type.num_fields):
            out.setdefault(b.type.field(i).name, []).extend(b.field(i).to_pylist())
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["test.fastq", "test.fastq.gz", "test_bgzip.fastq.gz"])
def test_gpu_fastq_batches_on_the_fixtures(ctx, name):
    """name / description / sequence / quality_scores (exon-fastq/src/array_builder.rs:68-102) of the reference's fixtures --
    plain, gzip (one member: gzip_stream.hip) and BGZF -- out of exon_hip_scan_bind_ctx + exon_hip_scan_next = the host reader
    = oracle/decode.py; slt/fastq-scan-test.slt:51-60 pins the row count (2)."""
    from oracle import decode
    p = os.path.join(FX, "fastq", name)
    recs = decode.decode_fastq(p)
    s = exon_amd.Scan(p, "fastq", batch_size=1, gpu_parse=True).bind_ctx(ctx)
    sch = s.schema()
    assert [sch.field(i).name for i in range(sch.num_fields)] == ["name", "description", "sequence", "quality_scores"]
    dev = _fastq_cols(s)
    assert s.decoded_on_gpu()[0]
    s.close()
    host = _fastq_cols(exon_amd.Scan(p, "fastq"))
    for k in ("name", "description", "sequence", "quality_scores"):
        assert dev[k] == host[k] == [r[k] for r in recs], k
    assert len(recs) == 2


@pytest.mark.gpu
def test_gpu_fastq_batches_equal_the_host_reader_on_synthetic_reads(ctx, tmp_path, monkeypatch):
    """300 k reads of ragged length with and without a description, CRLF line ends on a few, over several slabs (BGZF and plain
    text): device columns = host reader = oracle, read by read; the quality histogram of the same file still decodes on the device."""
    from oracle import decode
    rng = np.random.default_rng(17)
    n = 300_000
    lens = rng.integers(1, 180, n)
    bases = np.frombuffer(b"ACGTN", np.uint8)
    lines = []
    for i in range(n):
        L = int(lens[i])
        seq = bases[rng.integers(0, 5, L)].tobytes().decode()
        qual = (rng.integers(33, 74, L).astype(np.uint8)).tobytes().decode()
        head = f"@r{i}" + ("" if i % 3 == 0 else f" lane:{i % 7} x y" if i % 3 == 1 else " ")
        eol = "\r\n" if i % 1000 == 7 else "\n"
        lines.append(f"{head}{eol}{seq}{eol}+{eol}{qual}{eol}")
    p = tmp_path / "syn.fastq"
    p.write_text("".join(lines), newline="")
    gz = str(p) + ".gz"
    subprocess.check_call([BGZIP, str(p), gz, "6"])
    recs = decode.decode_fastq(str(p))
    assert len(recs) == n
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "8")
    for path in (gz, str(p)):
        s = exon_amd.Scan(path, "fastq", gpu_parse=True).bind_ctx(ctx)
        dev = _fastq_cols(s)
        assert s.decoded_on_gpu()[0], path
        s.close()
        for k in ("name", "description", "sequence", "quality_scores"):
            assert dev[k] == [r[k] for r in recs], (path, k)
    host = _fastq_cols(exon_amd.Scan(gz, "fastq"))
    assert host["description"] == dev["description"] and host["name"] == dev["name"]


@pytest.mark.gpu
def test_gpu_fastq_batches_hand_over_to_the_host_reader_mid_file(ctx, tmp_path, monkeypatch):
    """A .fastq.gz whose first member holds 60 000 reads and whose tail is 3000 one-read members: the device inflates the first
    slabs (1 MiB of text each), refuses the run of tiny members (more member ends in a chunk than it tracks), and the host reader
    takes over behind the last read emitted -- every read once, in file order; the scan says it was not decoded on the GPU."""
    import gzip
    def rec(i):
        return f"@n{i} d{i}\nACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII{chr(33 + i % 40)}\n"
    n1, n2 = 60_000, 3000
    p = tmp_path / "members.fastq.gz"
    with open(p, "wb") as f:
        f.write(gzip.compress("".join(rec(i) for i in range(n1)).encode(), 6))
        for i in range(n1, n1 + n2):
            f.write(gzip.compress(rec(i).encode(), 6))
    want = [dict(name=f"n{i}", description=f"d{i}", sequence="ACGTACGTACGTACGTACGTACGTACGTACGT", quality_scores="I" * 31 + chr(33 + i % 40)) for i in range(n1 + n2)]
    monkeypatch.setenv("EXON_HIP_GZ_SLAB_MB", "1")
    s = exon_amd.Scan(str(p), "fastq", gpu_parse=True).bind_ctx(ctx)
    dev = _fastq_cols(s)
    decoded = s.decoded_on_gpu()[0]
    s.close()
    assert len(dev["name"]) == n1 + n2
    for k in ("name", "description", "sequence", "quality_scores"):
        assert dev[k] == [r[k] for r in want], k
    assert not decoded


def test_region_batches_with_several_kept_runs_per_slab(ctx, tmp_path, monkeypatch):
    """A region that keeps several runs of rows inside one slab (blocks of contig 1 between blocks of contig 2; bam_region_filter's
    overlap test produces the same in front of a region's start): every run goes out as views (the row-by-row gather takes over
    beyond 256 runs per slab): GPU batches = host reader, with views and with the gather forced; a region nothing matches sends
    nothing."""
    head = "##fileformat=VCFv4.2\n##contig=<ID=1>\n##contig=<ID=2>\n##INFO=<ID=AF,Number=1,Type=Float,Description=\"a\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    rows, pos = [], 0
    for block, n in enumerate([5000, 3, 12000, 1, 7, 20000, 9000, 2, 15000]):
        chrom = "1" if block % 2 == 0 else "2"
        for _ in range(n):
            pos += 1
            rows.append(f"{chrom}\t{pos}\t{'.' if pos % 3 else 'rs' + str(pos)}\tA\tC\t{pos % 50}\tPASS\tAF=0.{1 + pos % 9}\n")
    p = tmp_path / "runs.vcf"
    p.write_text(head + "".join(rows))
    for region in ("1", "2", "1:5001-5010", "2:1-4"):
        want = _fastq_cols(exon_amd.Scan(str(p), "vcf", info_field="AF", region=region, project=("id",)))
        for forced in ("0", "1"):
            monkeypatch.setenv("EXON_HIP_EXPORT_GATHER", forced)
            s = exon_amd.Scan(str(p), "vcf", info_field="AF", region=region, gpu_parse=True, batch_size=1000, project=("id",)).bind_ctx(ctx)
            got = _fastq_cols(s)
            assert s.decoded_on_gpu()[0]
            s.close()
            assert got.keys() == want.keys(), (region, forced)
            for k in want:
                assert got[k] == want[k], (region, forced, k)
    assert len(_fastq_cols(exon_amd.Scan(str(p), "vcf", region="1")).get("pos", [])) == 5000 + 12000 + 7 + 9000 + 15000


def test_batches_outlive_the_scan(ctx, tmp_path):
    """Batches are views into pinned blocks shared by reference count (exon::BatchArena): closing the scan -- and dropping single
    columns of a batch early -- must not pull the memory from under the batches still held; their content is read afterwards."""
    import gc
    path = str(tmp_path / "t.vcf")
    subprocess.check_call([GEN, "vcf", "200000", path], stdout=subprocess.DEVNULL)
    want = _fastq_cols(exon_amd.Scan(path, "vcf", info_field="AF", project=("id", "ref", "alt")))
    s = exon_amd.Scan(path, "vcf", info_field="AF", gpu_parse=True, batch_size=5000, project=("id", "ref", "alt")).bind_ctx(ctx)
    held = list(s)
    s.close()
    del s
    cols_first = [b.field(0) for b in held[::2]]   # single columns kept, their batches dropped
    held = held[1::2]
    gc.collect()
    got = {}
    for b in held:
        for i in range(b.type.num_fields):
            got.setdefault(b.type.field(i).name, []).extend(b.field(i).to_pylist())
    n = 5000
    for k in want:
        expect = [v for j in range(1, 40, 2) for v in want[k][j * n:(j + 1) * n]]
        assert got[k] == expect, k
    chrom0 = [v for c in cols_first for v in c.to_pylist()]
    assert chrom0 == [v for j in range(0, 40, 2) for v in want["chrom"][j * n:(j + 1) * n]]
