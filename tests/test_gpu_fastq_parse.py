"""GPU-side FASTQ record splitting (exon_hip_fastq_parser_*) and K5 over views into the raw text, against the native
CPU decoder (exon_hip_scan_*) and the oracle's histogram: bit-exact counts."""
import os
import subprocess

import numpy as np
import pytest

import exon_amd
from oracle import decode

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
GEN = os.path.join(ROOT, "tools", "bin", "gen_text")


def oracle_hist(oracle, off, data, lmax):
    h, _ = oracle.c5_qual_pos_hist(off, data, lmax)
    return h.reshape(-1)


def views_hist(ctx, res, which, lmax):
    d_hist = ctx.zeros(np.int64, lmax * 256)
    v = res["views"]
    s, e = (v.seq_start, v.seq_end) if which == "seq" else (v.qual_start, v.qual_end)
    ctx.qual_pos_hist_views(res["d_text"], s, e, res["n_reads"], lmax, d_hist)
    ctx.sync()
    return d_hist.to_host()


def test_fastq_views_reference_fixture(ctx, oracle):
    path = os.path.join(FX, "fastq", "test.fastq")
    text = open(path, "rb").read()
    recs = decode.decode_fastq(path)
    p = exon_amd.FASTQParser(ctx, max_slab_bytes=1 << 20)
    res = p.parse_host(text)
    assert res["n_reads"] == len(recs) == 2 and res["n_undecided"] == 0 and res["consumed_bytes"] == len(text)
    for r, rec in enumerate(recs):
        assert text[res["seq_start"][r]:res["seq_end"][r]].decode() == rec["sequence"]
        assert text[res["qual_start"][r]:res["qual_end"][r]].decode() == rec["quality_scores"]
    off, data = decode.fastq_device_columns(recs)
    lmax = 512
    assert np.array_equal(views_hist(ctx, res, "qual", lmax), oracle_hist(oracle, off, data, lmax))
    p.close()


@pytest.mark.parametrize("ragged,crlf", [(False, False), (True, False), (True, True), (False, True)])
def test_fastq_views_synthetic_equals_oracle(ctx, oracle, ragged, crlf):
    rng = np.random.default_rng(11 + 2 * ragged + crlf)
    n, L = 20_000, 100
    nl = b"\r\n" if crlf else b"\n"
    quals, seqs, parts = [], [], []
    for i in range(n):
        ln = int(rng.integers(1, L + 1)) if ragged else L
        q = (rng.integers(33, 75, ln, dtype=np.uint8)).tobytes()
        s = rng.choice(np.frombuffer(b"ACGT", np.uint8), ln).tobytes()
        quals.append(q)
        seqs.append(s)
        parts += [b"@r%d desc" % i, nl, s, nl, b"+", nl, q, nl]
    text = b"".join(parts)
    p = exon_amd.FASTQParser(ctx, max_slab_bytes=len(text) + 1024)
    res = p.parse_host(text)
    assert res["n_reads"] == n and res["n_undecided"] == 0 and res["consumed_bytes"] == len(text)
    lens = np.array([len(q) for q in quals])
    assert np.array_equal(res["qual_end"] - res["qual_start"], lens)
    assert np.array_equal(res["seq_end"] - res["seq_start"], lens)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(lens)
    want_q = oracle_hist(oracle, off, np.frombuffer(b"".join(quals), np.uint8).copy(), 128)
    want_s = oracle_hist(oracle, off, np.frombuffer(b"".join(seqs), np.uint8).copy(), 128)
    assert np.array_equal(views_hist(ctx, res, "qual", 128), want_q)
    assert np.array_equal(views_hist(ctx, res, "seq", 128), want_s)
    p.close()


@pytest.mark.parametrize("misalign", [1, 5, 15])
def test_fastq_views_unaligned_text(ctx, oracle, misalign):
    """The slab may start anywhere (the BGZF pipeline puts the carried tail right in front of the inflated bytes)."""
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * (3 + i % 5), b"IHGF" * (3 + i % 5)) for i in range(3000))
    p = exon_amd.FASTQParser(ctx, max_slab_bytes=len(text) + 1024)
    a = p.parse_host(text)
    hist_a = views_hist(ctx, a, "qual", 64)  # the device views belong to the parser: use them before the next parse
    b = p.parse_host(text, misalign=misalign)
    assert b["n_reads"] == a["n_reads"] == 3000 and b["n_undecided"] == 0 and b["consumed_bytes"] == len(text)
    for k in ("seq_start", "seq_end", "qual_start", "qual_end"):
        assert np.array_equal(a[k], b[k])
    assert hist_a.sum() == sum(4 * (3 + i % 5) for i in range(3000))
    assert np.array_equal(hist_a, views_hist(ctx, b, "qual", 64))
    p.close()


def test_fastq_views_partial_slab_and_malformed(ctx):
    text = b"@a\nACGT\n+\nIIII\n@b\nAC\n+\nII\n@c\nACG"
    p = exon_amd.FASTQParser(ctx, max_slab_bytes=1 << 16)
    res = p.parse_host(text, final=False)  # the slab stops inside record c: two whole records are consumed
    assert res["n_reads"] == 2 and res["n_undecided"] == 0
    assert res["consumed_bytes"] == text.index(b"@c")
    res = p.parse_host(text + b"\n", final=True)  # ... but at end of input a partial record is not decidable here
    assert res["n_undecided"] > 0
    res = p.parse_host(b"@a\nACGT\nIIII\n+\n", final=True)  # '+' line missing
    assert res["n_undecided"] > 0
    res = p.parse_host(b"a\nACGT\n+\nIIII\n", final=True)  # no '@'
    assert res["n_undecided"] > 0
    res = p.parse_host(b"\n" * 4096, final=True)  # far more lines than records of >= 16 bytes allow
    assert res["n_undecided"] > 0
    p.close()


def _hist_through_scan(ctx, path, gpu_parse, lmax, column=3, compression=None, fallback=False):
    scan = exon_amd.Scan(str(path), "fastq", gpu_parse=gpu_parse, compression=compression)
    plan = ctx.plan_qual_pos_hist(lmax, columns=(column,))
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    st.close()
    plan.close()
    assert scan.decoded_on_gpu()[0] == (bool(gpu_parse) and not fallback), 'silent host fallback'
    scan.close()
    return rows, np.array(counts)


@pytest.mark.parametrize("ragged", [0, 1])
def test_fastq_file_to_gpu_pipeline_equals_host_decode(ctx, tmp_path, monkeypatch, ragged):
    n = 300_000
    path = tmp_path / "syn.fastq"
    subprocess.check_call([GEN, "fastq", str(n), str(path), "150", str(ragged)])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "8")  # many slabs: records straddle slab boundaries
    rows_g, gpu = _hist_through_scan(ctx, path, True, 256)
    rows_h, host = _hist_through_scan(ctx, path, False, 256)
    assert rows_g == rows_h == n
    assert np.array_equal(gpu, host)
    assert gpu.sum() == host.sum() and gpu.sum() > n * 100
    # the sequence lines through the same machinery
    rows_g, gpu = _hist_through_scan(ctx, path, True, 256, column=2)
    rows_h, host = _hist_through_scan(ctx, path, False, 256, column=2)
    assert rows_g == rows_h == n and np.array_equal(gpu, host)


@pytest.mark.parametrize("ragged", [0, 1])
def test_fastq_bgzf_is_inflated_on_the_gpu(ctx, tmp_path, monkeypatch, ragged):
    n = 200_000
    path = tmp_path / "syn.fastq"
    subprocess.check_call([GEN, "fastq", str(n), str(path), "150", str(ragged)])
    gz = tmp_path / "syn.fastq.gz"
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "bgzip"), str(path), str(gz), "6"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "4")
    rows_g, gpu = _hist_through_scan(ctx, gz, True, 256)
    rows_p, plain = _hist_through_scan(ctx, path, True, 256)
    rows_h, host = _hist_through_scan(ctx, gz, False, 256)
    assert rows_g == rows_p == rows_h == n
    assert np.array_equal(gpu, plain) and np.array_equal(gpu, host)


def test_fastq_gpu_pipeline_reference_fixtures(ctx, oracle):
    for name, comp in (("test.fastq", None), ("test_bgzip.fastq.gz", "gzip")):
        path = os.path.join(FX, "fastq", name)
        rows, gpu = _hist_through_scan(ctx, path, True, 512, compression=comp)
        off, data = decode.fastq_device_columns(decode.decode_fastq(path))
        assert rows == 2 and np.array_equal(gpu, oracle_hist(oracle, off, data, 512))


def test_fastq_gpu_pipeline_falls_back_to_host(ctx, tmp_path):
    """Blank lines between records are legal for the host reader but not decidable by the line-index split: the state is
    restored and the file is decoded on the host."""
    path = tmp_path / "blank.fastq"
    with open(path, "wb") as f:
        for i in range(1000):
            f.write(b"@r%d\nACGTACGT\n+\nIIIIHHHH\n" % i)
            if i == 500:
                f.write(b"\n")
    rows_g, gpu = _hist_through_scan(ctx, path, True, 64, fallback=True)
    rows_h, host = _hist_through_scan(ctx, path, False, 64)
    assert rows_g == rows_h == 1000 and np.array_equal(gpu, host)
    assert gpu[0 * 256 + ord("I")] == 1000 and gpu[7 * 256 + ord("H")] == 1000
