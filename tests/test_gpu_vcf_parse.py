"""GPU-side VCF record parsing (exon_hip_vcf_parser_*) against the ORACLE's decoder (oracle/decode.py, which restates the
reference's builders) and, as a second opinion, the product's native CPU decoder: bit-identical columns, same dictionaries
(up to id permutation for FILTER, resolved through the names)."""
import os
import subprocess

import numpy as np
import pytest

import exon_amd
from oracle_expect import k4_expected, vcf_columns

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")


def bits(bm, n):
    return np.unpackbits(bm, bitorder="little")[:n].astype(bool)


def data_lines(path):
    raw = open(path, "rb").read()
    lines = [ln for ln in raw.split(b"\n") if ln and not ln.startswith(b"#")]
    return b"\n".join(lines) + b"\n"


def cpu_columns(path, info_field=None, monkeypatch=None):
    s = exon_amd.Scan(path, "vcf", info_field=info_field)
    batches = list(s)
    out = {"chrom": [x for b in batches for x in b.field(0).to_pylist()],
           "pos": [x for b in batches for x in b.field(1).to_pylist()],
           "qual": [x for b in batches for x in b.field(2).to_pylist()],
           "filter": [x for b in batches for x in b.field(3).to_pylist()],
           "contigs": s.dictionary(0)}
    if info_field:
        out["info"] = [x for b in batches for x in b.field(4).to_pylist()]
    s.close()
    return out


def check(res, cpu, contigs, filters, info=False):
    n = res["n_rows"]
    assert n == len(cpu["chrom"]) and res["n_undecided"] == 0
    assert [contigs[i] for i in res["chrom_id"]] == cpu["chrom"]
    pv, qv = bits(res["pos_valid"], n), bits(res["qual_valid"], n)
    assert [int(p) if v else None for p, v in zip(res["pos"], pv)] == cpu["pos"]
    want_q = np.array([np.float32(x) if x is not None else np.float32(0) for x in cpu["qual"]], np.float32)
    assert np.array_equal(qv, np.array([x is not None for x in cpu["qual"]]))
    assert np.array_equal(res["qual"][qv].view(np.uint32), want_q[qv].view(np.uint32))
    assert [filters[i] for i in res["filter_id"]] == cpu["filter"]
    if info:
        iv = bits(res["info_valid"], n)
        want = np.array([np.float32(x) if x is not None else np.float32(0) for x in cpu["info"]], np.float32)
        assert np.array_equal(iv, np.array([x is not None for x in cpu["info"]]))
        assert np.array_equal(res["info"][iv].view(np.uint32), want[iv].view(np.uint32))


def test_gpu_parse_reference_fixture(ctx):
    path = os.path.join(FX, "vcf", "index.vcf")
    orc = vcf_columns(path, "vcf", "MQ0F")                     # the oracle's decoder: the parity reference
    cpu = cpu_columns(path, "MQ0F")                            # the product's host decoder: second opinion
    p = exon_amd.VCFParser(ctx, orc["contigs"], info_field="MQ0F")
    res = p.parse_host(data_lines(path))
    check(res, orc, orc["contigs"], p.filters(), info=True)
    check(res, cpu, cpu["contigs"], p.filters(), info=True)
    assert res["n_rows"] == 621  # slt/vcf-select-tests.slt:47-50
    p.close()


def test_gpu_parse_synthetic_slabs_and_filter_dictionary(ctx, tmp_path, oracle):
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    if not os.path.exists(gen):
        subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "gen_text.cpp"), "-o", gen])
    n = 300_000
    path = tmp_path / "syn.vcf"
    subprocess.check_call([gen, "vcf", str(n), str(path)])
    cpu = cpu_columns(path, "AF")
    orc = vcf_columns(path, "vcf", "AF")
    text = data_lines(path)
    p = exon_amd.VCFParser(ctx, orc["contigs"], info_field="AF", max_slab_bytes=8 << 20)
    # two slabs cut at a line boundary: the FILTER dictionary persists across slabs
    cut = text.rfind(b"\n", 0, len(text) // 2) + 1
    r1, r2 = p.parse_host(text[:cut]), p.parse_host(text[cut:])
    filters = p.filters()
    assert sorted(filters) == sorted(set(cpu["filter"]))
    res = {k: (np.concatenate([r1[k], r2[k]]) if k in ("chrom_id", "pos", "qual", "filter_id", "info") else None) for k in r1}
    n1 = r1["n_rows"]
    assert n1 % 8 != 0 or True
    for k in ("pos_valid", "qual_valid", "info_valid"):
        res[k] = np.packbits(np.concatenate([bits(r1[k], n1), bits(r2[k], r2["n_rows"])]), bitorder="little")
    res["n_rows"], res["n_undecided"] = n1 + r2["n_rows"], r1["n_undecided"] + r2["n_undecided"]
    check(res, orc, orc["contigs"], filters, info=True)
    check(res, cpu, cpu["contigs"], filters, info=True)
    # and the parsed AF column equals the generator's (text round trip through the GPU parser)
    af, av, q, qv, fid = oracle.gen_c4(4, 0, n)
    avb = bits(av, n)
    assert np.array_equal(bits(res["info_valid"], n), avb)
    assert np.array_equal(res["info"][avb].view(np.uint32), af[avb].view(np.uint32))
    p.close()


@pytest.mark.parametrize("misalign", [1, 7, 15])
def test_gpu_parse_unaligned_text(ctx, misalign):
    path = os.path.join(FX, "vcf", "index.vcf")
    cpu = cpu_columns(path, "MQ0F")
    text = data_lines(path)
    p = exon_amd.VCFParser(ctx, cpu["contigs"], info_field="MQ0F")
    a = p.parse_host(text)
    b = p.parse_host(text + b"1\t5\t.", misalign=misalign)  # plus a trailing partial line
    assert b["n_rows"] == a["n_rows"] == 621 and b["n_undecided"] == 0
    for k in ("chrom_id", "pos", "qual", "filter_id", "info"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    p.close()


def test_gpu_parse_undecidable_rows_are_counted(ctx):
    contigs = ["1", "2"]
    text = (b"1\t5\t.\tA\tC\t30.5\tPASS\tDP=3;AF=0.5\n"
            b"3\t6\t.\tA\tC\t1\tPASS\tAF=0.1\n"                                     # contig not in the header
            b"2\t7\trs1\tA\tC\t0.12345678901234567890123\tq10\tAF=0.25\n"          # > 19 significant digits
            b"2\t0\t.\tA\tC\t.\t.\t.\n"                                              # pos 0 -> NULL, qual NULL, [] filter, NULL info
            b"2\t9\t.\tA\tC\t7e-1\tq10;s50\tDP=1;AF=.;X\n")
    p = exon_amd.VCFParser(ctx, contigs, info_field="AF")
    res = p.parse_host(text)
    assert res["n_rows"] == 5 and res["n_undecided"] == 2
    f = p.filters()
    assert [f[i] for i in res["filter_id"]] == ["PASS", "PASS", "q10", "", "q10;s50"]
    assert bits(res["pos_valid"], 5).tolist() == [True, True, True, False, True]
    assert bits(res["qual_valid"], 5).tolist() == [True, True, False, False, True]
    assert bits(res["info_valid"], 5).tolist() == [True, True, True, False, False]
    assert res["qual"][4] == np.float32(0.7) and res["info"][0] == np.float32(0.5)
    p.close()


def test_gpu_parse_pos_takes_one_leading_plus(ctx):
    """POS through Rust's usize::from_str (tests/test_decoder_edge_cases.py has the evidence): "+5" is 5 and "+0" NULL ON THE
    DEVICE, no undecided row; "++5", "+" and "-5" are not numbers: undecided, the host reports them."""
    text = (b"1\t+5\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t+0\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t+0012\t.\tA\tC\t1\tPASS\tAF=0.5\n")
    p = exon_amd.VCFParser(ctx, ["1"], info_field="AF")
    res = p.parse_host(text)
    assert res["n_rows"] == 3 and res["n_undecided"] == 0
    assert res["pos"].tolist()[0] == 5 and res["pos"].tolist()[2] == 12 and bits(res["pos_valid"], 3).tolist() == [True, False, True]
    res = p.parse_host(b"1\t++5\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t+\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t-5\t.\tA\tC\t1\tPASS\tAF=0.5\n")
    assert res["n_rows"] == 3 and res["n_undecided"] == 3
    p.close()


def test_gpu_decimal_to_f32_is_the_nearest_value_bit_for_bit(ctx):
    """QUAL and a Float INFO value over every shape of plain decimal: short significands with up to ten fraction digits (one exact
    IEEE division on the device: Clinger's case), long ones (Eisel-Lemire), leading zeros, integers, values around 2^24.  The bits
    must be those of the correctly rounded binary32 -- glibc's strtof -- which is what Rust's f32::from_str (noodles' QUAL and INFO
    Float) gives."""
    rng = np.random.default_rng(11)
    vals = ["0", "0.0", "1", "16777215", "16777216", "16777217", "0.1", "0.0000000001", "123.4567", "9999999", "0.9999999", "381.1",
            "0.000684898929", "33554433", "1.0000001", "0.30000001192092896", "8388607.5", "0.0000001234567", "7.000000", "000012.50"]
    for _ in range(20000):
        nd = int(rng.integers(1, 13))
        digits = "".join(str(int(d)) for d in rng.integers(0, 10, nd))
        k = int(rng.integers(0, nd + 1))
        vals.append((digits[:k] or "0") + "." + (digits[k:] or "0") if rng.random() < 0.8 else digits)
        if rng.random() < 0.2:
            vals[-1] = "0." + "0" * int(rng.integers(0, 9)) + digits
    text = "".join(f"1\t{i + 1}\t.\tA\tC\t{v}\tPASS\tAF={vals[-1 - i]}\n" for i, v in enumerate(vals)).encode()
    p = exon_amd.VCFParser(ctx, ["1"], info_field="AF", max_slab_bytes=len(text) + 4096)
    res = p.parse_host(text)
    n = len(vals)
    assert res["n_rows"] == n and res["n_undecided"] == 0
    import ctypes
    libc = ctypes.CDLL(None)  # glibc's strtof rounds the decimal to binary32 directly (numpy's float32(str) goes through a double)
    libc.strtof.restype = ctypes.c_float
    libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    want_q = np.array([libc.strtof(v.encode(), None) for v in vals], np.float32)
    want_i = want_q[::-1]
    assert np.array_equal(res["qual"].view(np.uint32), want_q.view(np.uint32))
    assert np.array_equal(res["info"].view(np.uint32), want_i.view(np.uint32))
    p.close()


def _k4_through_scan(ctx, path, gpu_parse, info_field="AF", fallback=False, thr=0.01):
    scan = exon_amd.Scan(path, "vcf", info_field=info_field, gpu_parse=gpu_parse)
    plan = ctx.plan_cmp_avg_by_group(">", thr, 64, columns=(4, 2, 3))
    st = plan.open()
    rows = st.consume(scan)
    counts, sums = st.finish()
    names = scan.dictionary(3)
    res = {names[g]: (int(counts[g]), int(counts[64 + g]), float(sums[g])) for g in range(len(names)) if counts[64 + g]}
    st.close()
    plan.close()
    assert scan.decoded_on_gpu()[0] == (bool(gpu_parse) and not fallback), 'silent host fallback'
    scan.close()
    return rows, res


def _same(a, b, rel=1e-9):
    assert a.keys() == b.keys()
    for k in b:
        assert a[k][:2] == b[k][:2], k                     # counts bit-exact
        assert a[k][2] == pytest.approx(b[k][2], rel=rel), k


@pytest.mark.parametrize("kind", ["text", "bgzf"])
def test_file_to_gpu_pipeline_equals_the_oracle(ctx, oracle, tmp_path, monkeypatch, kind):
    """file -> (GPU inflate) -> GPU parse -> K4 against oracle/decode.py + oracle/exon_oracle.c over the same file."""
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    n = 250_000
    path = tmp_path / "syn.vcf"
    subprocess.check_call([gen, "vcf", str(n), str(path)])
    rows_o, want = k4_expected(oracle, path, "vcf", "AF")
    if kind == "bgzf":
        gz = tmp_path / "syn.vcf.gz"
        subprocess.check_call([os.path.join(ROOT, "tools", "bin", "bgzip"), str(path), str(gz), "6"])
        path = gz
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "4")
    rows_g, gpu = _k4_through_scan(ctx, path, True)
    assert rows_g == rows_o == n
    _same(gpu, want)
    # the reference fixture too (typed INFO field MQ0F, all FILTER lists empty)
    fx = os.path.join(FX, "vcf", "index.vcf.gz" if kind == "bgzf" else "index.vcf")
    rows_o, want = k4_expected(oracle, fx, "vcf", "MQ0F", thr=-1.0)
    rows_g, gpu = _k4_through_scan(ctx, fx, True, info_field="MQ0F", thr=-1.0)
    assert rows_g == rows_o == 621
    _same(gpu, want)


def test_file_to_gpu_parse_pipeline_equals_host_decode(ctx, tmp_path, monkeypatch):
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    n = 2_000_000
    path = tmp_path / "syn.vcf"
    subprocess.check_call([gen, "vcf", str(n), str(path)])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "16")  # several slabs, double-buffered reader
    rows_g, gpu = _k4_through_scan(ctx, path, True)
    rows_h, host = _k4_through_scan(ctx, path, False)
    assert rows_g == rows_h == n
    assert gpu.keys() == host.keys()
    for k in host:
        assert gpu[k][:2] == host[k][:2]                      # counts bit-exact
        assert gpu[k][2] == pytest.approx(host[k][2], rel=1e-12)


@pytest.mark.parametrize("slab_mb", ["4", "64"])
def test_bgzf_file_is_inflated_and_parsed_on_the_gpu(ctx, tmp_path, monkeypatch, slab_mb):
    """file.vcf.gz (BGZF): compressed blocks -> HBM -> GPU inflate -> GPU parse -> K4, equal to the host-inflate + host-decode
    path and to the plain-text twin; lines straddle BGZF blocks and slabs (carried device-to-device)."""
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    bgzip = os.path.join(ROOT, "tools", "bin", "bgzip")
    n = 1_500_000
    path = tmp_path / "syn.vcf"
    subprocess.check_call([gen, "vcf", str(n), str(path)])
    gz = tmp_path / "syn.vcf.gz"
    subprocess.check_call([bgzip, str(path), str(gz), "6"])
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", slab_mb)
    rows_g, gpu = _k4_through_scan(ctx, gz, True)
    rows_p, plain = _k4_through_scan(ctx, path, True)
    monkeypatch.setenv("EXON_HIP_GPU_INFLATE", "0")  # host threads inflate, GPU parses
    rows_i, hostinf = _k4_through_scan(ctx, gz, True)
    rows_h, host = _k4_through_scan(ctx, gz, False)
    assert rows_g == rows_p == rows_i == rows_h == n
    for other in (plain, hostinf, host):
        assert gpu.keys() == other.keys()
        for k in other:
            assert gpu[k][:2] == other[k][:2]
            assert gpu[k][2] == pytest.approx(other[k][2], rel=1e-12)


def test_bgzf_reference_fixture_through_the_gpu(ctx):
    """index.vcf.gz of the reference (621 records) through GPU inflate + GPU parse."""
    path = os.path.join(FX, "vcf", "index.vcf.gz")
    scan = exon_amd.Scan(path, "vcf", gpu_parse=True)
    plan = ctx.plan_region_count(scan.dictionary(0).index("1"), 1, None, columns=(0, 1))
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    st.close(); plan.close(); scan.close()
    assert rows == 621 and int(counts[0]) == 191  # exon-core/src/session_context/exon_context_ext.rs / slt pins


def test_corrupt_bgzf_block_falls_back_to_the_host_error(ctx, tmp_path):
    """A flipped byte inside a block: the device reports the block, the host decoder is asked instead and raises."""
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    bgzip = os.path.join(ROOT, "tools", "bin", "bgzip")
    path = tmp_path / "c.vcf"
    subprocess.check_call([gen, "vcf", "300000", str(path)])
    gz = tmp_path / "c.vcf.gz"
    subprocess.check_call([bgzip, str(path), str(gz), "6"])
    raw = bytearray(open(gz, "rb").read())
    raw[len(raw) // 2] ^= 0x5A
    open(gz, "wb").write(bytes(raw))
    with pytest.raises(exon_amd.ExonHipError):
        _k4_through_scan(ctx, gz, True)


def test_gpu_parse_falls_back_to_host_on_undecidable_rows(ctx, tmp_path):
    """A contig that is not in the header makes the device give up on the slab: the state is restored and the file is
    re-decoded on the host, so the answer is still the host decoder's."""
    path = tmp_path / "odd.vcf"
    with open(path, "w") as f:
        f.write('##fileformat=VCFv4.3\n##contig=<ID=1>\n##INFO=<ID=AF,Number=1,Type=Float,Description="AF">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for i in range(5000):
            chrom = "GL99" if i == 4000 else "1"
            f.write(f"{chrom}\t{i + 1}\t.\tA\tC\t{i % 90}.5\t{'PASS' if i % 3 else 'q10'}\tAF=0.{1 + i % 8}\n")
    rows_g, gpu = _k4_through_scan(ctx, path, True, fallback=True)
    rows_h, host = _k4_through_scan(ctx, path, False)
    assert rows_g == rows_h == 5000 and gpu == host


def test_strict_mode_turns_the_host_fallback_into_an_error(ctx, tmp_path, monkeypatch):
    """EXON_HIP_GPU_PARSE_STRICT=1: a deployment that must never decode on the host gets an error instead of the fallback."""
    path = tmp_path / "odd.vcf"
    with open(path, "w") as f:
        f.write('##fileformat=VCFv4.3\n##contig=<ID=1>\n##INFO=<ID=AF,Number=1,Type=Float,Description="AF">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for i in range(500):
            f.write(f"{'GL99' if i == 400 else '1'}\t{i + 1}\t.\tA\tC\t1.5\tPASS\tAF=0.5\n")
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_STRICT", "1")
    scan = exon_amd.Scan(path, "vcf", info_field="AF", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
    st = plan.open()
    with pytest.raises(exon_amd.ExonHipError, match="EXON_HIP_GPU_PARSE_STRICT"):
        st.consume(scan)
    st.close()
    plan.close()
    scan.close()


@pytest.mark.parametrize("fixture", ["index.vcf", "index.vcf.gz"])
@pytest.mark.parametrize("field,thr", [("DP", 0), ("DP", 1), ("DP", 4), ("MQ0F", -1.0), ("MQSB", 0.5), ("SGB", -0.5)])
def test_fixture_aggregates_against_a_regex_statement_of_the_query(ctx, fixture, field, thr):
    """A second opinion that shares no code with the oracle or the decoders: `SELECT filter, COUNT(qual), COUNT(*), SUM(qual)
    WHERE info.<field> > thr GROUP BY filter` over the reference's 621-row fixture, stated with `str.split` and one regular
    expression on the text (an Integer field compared as integers, a Float field through float32, as the reference types
    them: exon-core/src/datasources/vcf/schema_builder.rs:197-205), against file -> (GPU inflate) -> GPU parse -> K4."""
    import gzip
    import re
    path = os.path.join(FX, "vcf", fixture)
    text = (gzip.open(path, "rt") if fixture.endswith(".gz") else open(path)).read()
    want = {}
    pat = re.compile(r"(?:^|;)" + field + r"=([^;]+)")
    n = 0
    for line in text.split("\n"):
        if not line or line.startswith("#"):
            continue
        n += 1
        c = line.split("\t")
        m = pat.search(c[7])
        if not m:
            continue  # NULL never passes a comparison
        x = int(m.group(1)) if field in ("DP", "IDV") else float(np.float32(float(m.group(1))))
        if not x > thr:
            continue
        key = "" if c[6] == "." else c[6]
        cnt_q, cnt, sm = want.get(key, (0, 0, 0.0))
        if c[5] != ".":
            cnt_q, sm = cnt_q + 1, sm + float(np.float32(float(c[5])))
        want[key] = (cnt_q, cnt + 1, sm)
    assert n == 621 and want, "the fixture and the field must give the query something to do"
    rows, got = _k4_through_scan(ctx, path, True, info_field=field, thr=thr)
    assert rows == 621
    _same(got, want)
