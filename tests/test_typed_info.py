"""Typed INFO fields (InfosBuilder, exon-vcf/src/array_builder/info_builder.rs:152-309 + typing in
exon-core/src/datasources/vcf/schema_builder.rs:197-249): several keys per scan, Float -> f32, Integer -> i32 (exact), Flag ->
Boolean (true when present, NULL when absent), String / Character -> dictionary, whole struct NULL when INFO is '.'.  CPU tests compare
the host decoders with the oracle's decoder; the gpu-marked ones run the same expectations through the device parsers."""
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

HEAD = ('##fileformat=VCFv4.3\n##contig=<ID=1>\n'
        '##INFO=<ID=AF,Number=1,Type=Float,Description="x">\n##INFO=<ID=DP,Number=1,Type=Integer,Description="x">\n'
        '##INFO=<ID=DB,Number=0,Type=Flag,Description="x">\n##INFO=<ID=CSQ,Number=1,Type=String,Description="x">\n'
        '##INFO=<ID=AA,Number=1,Type=Character,Description="x">\n##INFO=<ID=AC,Number=A,Type=Integer,Description="x">\n'
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
ROWS = ["AF=0.5;DP=10;DB;CSQ=missense;AA=T", "DP=7;AF=.;CSQ=.", ".", "DB;DBX=1;XAF=3;AF=0.25", "CSQ=stop;AF=1e-3;AF=0.9;DP=3", "DP;DB=1;AA=.",
        "AC=1,2;CSQ=missense"]


def oracle_typed(v, fields):
    """per field: python values (int / float / True / str / list / None) typed by the oracle from the header
    (decode.typed_info: schema_builder.rs:197-249 + info_builder.rs:152-309)"""
    return [decode.typed_info(v, f)[1] for f in fields]


def host_typed(path, fmt, fields):
    s = exon_amd.Scan(path, fmt, info_field=",".join(fields))
    batches = list(s)
    cols = [[x for b in batches for x in b.field(4 + k).to_pylist()] for k in range(len(fields))]
    types = [str(batches[0].type.field(4 + k).type) for k in range(len(fields))]
    s.close()
    return cols, types


def test_host_decoder_typed_info_kinds_and_null_rules(tmp_path):
    p = tmp_path / "t.vcf"
    p.write_text(HEAD + "".join(f"1\t{i + 1}\t.\tA\tC\t1\tPASS\t{r}\n" for i, r in enumerate(ROWS)))
    fields = ["AF", "DP", "DB", "CSQ"]
    got, types = host_typed(p, "vcf", fields)
    assert types == ["float", "int32", "bool", "dictionary<values=string, indices=int32, ordered=0>"]
    assert got[1] == [10, 7, None, None, 3, None, None] and all(isinstance(x, int) for x in got[1] if x is not None)
    want = oracle_typed(decode.decode_vcf(str(p)), fields)
    assert got == want
    assert got[0] == [0.5, None, None, 0.25, pytest.approx(1e-3), None, None]        # first occurrence wins; '.' -> NULL
    assert got[2] == [True, None, None, True, None, True, None]                        # Flag: true when present, NULL when absent
    assert got[3] == ["missense", None, None, None, "stop", None, "missense"]
    assert host_typed(p, "vcf", ["AA"])[0] == [["T", None, None, None, None, None, None]]
    ac, ac_type = host_typed(p, "vcf", ["AC"])                                          # Number=A -> List<Int32>
    assert ac_type == ["list<item: int32>"] and ac[0] == [None, None, None, None, None, None, [1, 2]]
    assert ac == oracle_typed(decode.decode_vcf(str(p)), ["AC"])
    with pytest.raises(exon_amd.ExonHipError, match="NOPE"):
        host_typed(p, "vcf", ["NOPE"])
    assert len(host_typed(p, "vcf", ["AF", "DP", "DB", "CSQ", "AA", "AC"])[0]) == 6     # more than round 2's four keys
    with pytest.raises(exon_amd.ExonHipError, match="at most 16"):
        host_typed(p, "vcf", ["AF"] * 17)
    bad = tmp_path / "bad.vcf"
    for val in ("1.5", "2147483648", "x", "1,2"):                                       # not an int32: the record's parse error
        bad.write_text(HEAD + f"1\t1\t.\tA\tC\t1\tPASS\tDP={val}\n")
        with pytest.raises(exon_amd.ExonHipError, match="integer"):
            host_typed(bad, "vcf", ["DP"])
        with pytest.raises(ValueError):
            decode.typed_info(decode.decode_vcf(str(bad)), "DP")


def test_every_info_kind_vcf_and_bcf_twin(tmp_path):
    """Float / Integer / Flag / String scalars and Integer / Float / String lists, written by tests/vcf_bcf_writer.py as a VCF
    and its BCF twin: host decoders == the rows they were written from == the oracle's typing of its own decode."""
    import vcf_bcf_writer as W
    rows = W.make_rows(3000)
    vcf, bcf = tmp_path / "k.vcf", tmp_path / "k.bcf"
    W.write_vcf(vcf, rows)
    W.write_bcf(bcf, rows, BGZIP)
    fields = [n for n, _, _ in W.INFO_HEADER]
    want = [W.expected_column(rows, f) for f in fields]
    want_types = ["float", "int32", "bool", "dictionary<values=string, indices=int32, ordered=0>", "list<item: int32>",
                  "list<item: float>", "list<item: dictionary<values=string, indices=int32, ordered=0>>"]
    assert sum(x is not None for x in want[4]) > 1000 and any(None in x for x in want[4] if x)
    for path, fmt, dec in ((vcf, "vcf", decode.decode_vcf), (bcf, "bcf", decode.decode_bcf)):
        got, types = host_typed(path, fmt, fields)
        assert types == want_types, fmt
        for k, f in enumerate(fields):
            assert got[k] == want[k], (fmt, f)
        o = dec(str(path))
        for k, f in enumerate(fields):
            assert decode.typed_info(o, f)[1] == want[k], (fmt, f, "oracle")


def test_reference_fixture_typed_info_vcf_and_bcf_twin():
    fields = ["DP", "MQ0F", "INDEL", "IDV"]
    v = decode.decode_vcf(os.path.join(FX, "vcf", "index.vcf"))
    want = oracle_typed(v, fields)
    got, types = host_typed(os.path.join(FX, "vcf", "index.vcf"), "vcf", fields)
    assert types == ["int32", "float", "bool", "int32"] and got == want
    assert sum(x is not None for x in want[0]) == 621 and all(x is None for x in want[2])
    gotb, typesb = host_typed(os.path.join(FX, "bcf", "index.bcf"), "bcf", fields)
    assert typesb == types and gotb == want
    assert oracle_typed(decode.decode_bcf(os.path.join(FX, "bcf", "index.bcf")), fields) == want


def test_big_file_parallel_reader_typed_info(tmp_path):
    """> 8 MiB: the multi-threaded slab reader carries numeric + Flag fields; a String field keeps the reader sequential."""
    n = 300_000
    p = tmp_path / "s.vcf"
    subprocess.check_call([GEN, "vcf", str(n), str(p)])
    assert os.path.getsize(p) > (8 << 20)
    got, _ = host_typed(p, "vcf", ["AF"])
    want = oracle_typed(decode.decode_vcf(str(p)), ["AF"])
    assert got == want and sum(x is None for x in want[0]) > 1000


# ---- GPU -----------------------------------------------------------------------------------------------------------------
def _bits(bm, n):
    return np.unpackbits(bm, bitorder="little")[:n].astype(bool)


@pytest.mark.gpu
def test_gpu_vcf_parser_typed_info(ctx, tmp_path):
    p = tmp_path / "t.vcf"
    rows = [r for r in ROWS if "1e-3" not in r or True]
    p.write_text(HEAD + "".join(f"1\t{i + 1}\t.\tA\tC\t1\tPASS\t{r}\n" for i, r in enumerate(rows)))
    want = oracle_typed(decode.decode_vcf(str(p)), ["AF", "DP", "DB"])
    text = "".join(f"1\t{i + 1}\t.\tA\tC\t1\tPASS\t{r}\n" for i, r in enumerate(rows)).encode()
    par = exon_amd.VCFParser(ctx, ["1"], info_field="AF,DP:i,DB:b")
    d = ctx.to_device(np.frombuffer(text + bytes(64), np.uint8))
    cols = par.parse_device(d.ptr, len(text))
    n = cols.n_rows
    assert n == len(rows) and cols.n_undecided == 0 and cols.n_info == 3

    def dev(ptr, dtype, count):
        out = np.empty(count, dtype)
        ctx._check(ctx.lib.exon_hip_memcpy_d2h(ctx.h, out.ctypes.data, ptr, out.nbytes, None))
        return out
    for k in range(3):
        valid = _bits(dev(cols.infos_valid[k], np.uint8, (n + 7) // 8), n)
        if want[k] and isinstance(next((x for x in want[k] if x is not None), None), bool):
            assert not cols.infos[k] and valid.tolist() == [x is True for x in want[k]]
        else:
            vals = dev(cols.infos[k], np.int32 if k == 1 else np.float32, n)  # DP is Type=Integer: an Int32 column
            assert valid.tolist() == [x is not None for x in want[k]]
            assert [v.item() for v, ok in zip(vals, valid) if ok] == [x for x in want[k] if x is not None]
    par.close()


@pytest.mark.gpu
def test_gpu_vcf_parser_list_valued_info(ctx, tmp_path):
    """List<Int32> / List<Float32> INFO fields decoded on the device (Arrow List layout: list validity, int32 offsets, items,
    item validity) == the rows the file was written from; Number=1 Integer next to them stays an exact Int32 column."""
    import vcf_bcf_writer as W
    rows = W.make_rows(20_000, seed=5)
    path = tmp_path / "l.vcf"
    W.write_vcf(path, rows)
    raw = open(path, "rb").read()
    body = raw[raw.index(b"#CHROM"):]
    text = body[body.index(b"\n") + 1:]
    par = exon_amd.VCFParser(ctx, ["1", "2"], info_field="AC:I,MQS:F,DP:i", max_slab_bytes=len(text) + 4096)
    d = ctx.to_device(np.frombuffer(text + bytes(64), np.uint8))
    cols = par.parse_device(d.ptr, len(text))
    n = cols.n_rows
    assert n == len(rows) and cols.n_undecided == 0 and cols.n_info == 3 and cols.info_kinds[:3] == b"IFi"

    def dev(ptr, dtype, count):
        out = np.empty(count, dtype)
        if count:
            ctx._check(ctx.lib.exon_hip_memcpy_d2h(ctx.h, out.ctypes.data, ptr, out.nbytes, None))
        return out
    for k, (key, dtype) in enumerate((("AC", np.int32), ("MQS", np.float32))):
        want = W.expected_column(rows, key)
        valid = _bits(dev(cols.infos_valid[k], np.uint8, (n + 7) // 8), n)
        off = dev(cols.list_offsets[k], np.int32, n + 1)
        total = int(off[-1])
        items = dev(cols.infos[k], dtype, total)
        ivalid = _bits(dev(cols.list_item_valid[k], np.uint8, (total + 7) // 8), total) if total else np.zeros(0, bool)
        assert off[0] == 0 and np.all(np.diff(off) >= 0) and total == sum(len(x) for x in want if x is not None)
        got = [None if not valid[r] else [items[i].item() if ivalid[i] else None for i in range(off[r], off[r + 1])] for r in range(n)]
        assert got == want, key
    dp = W.expected_column(rows, "DP")
    v = _bits(dev(cols.infos_valid[2], np.uint8, (n + 7) // 8), n)
    vals = dev(cols.infos[2], np.int32, n)
    assert [vals[r].item() if v[r] else None for r in range(n)] == dp
    par.close()


@pytest.mark.gpu
def test_gpu_vcf_parser_comma_dense_list_is_handed_back_not_overrun(ctx):
    """`AF=,,,,` is legal (every item is an EMPTY = NULL item) and holds one item per byte: more items than the device buffers,
    sized for slab bytes / 2 + 1, can take.  The parser must report the slab as undecided (host decoder) WITHOUT writing past
    its item-flag / bitmap buffers: a guard allocation made right after the parser's keeps its pattern, and a normal slab
    parsed next on the same parser still decodes."""
    lines = [b"1\t%d\t.\tA\tC\t1\tPASS\tAF=" % (i + 1) + b"," * 4000 + b"\n" for i in range(64)]
    text = b"".join(lines)
    cap = len(text) + 4096
    par = exon_amd.VCFParser(ctx, ["1"], info_field="AF:F", max_slab_bytes=cap)
    guard = ctx.to_device(np.full(1 << 20, 0xA5, np.uint8))
    d = ctx.to_device(np.frombuffer(text + bytes(64), np.uint8))
    cols = par.parse_device(d.ptr, len(text))
    assert cols.n_rows == 64 and cols.n_undecided > 0                    # 64 x 4001 items > cap / 2 + 1
    assert np.all(guard.to_host() == 0xA5)
    few = b"".join(b"1\t%d\t.\tA\tC\t1\tPASS\tAF=,0.5,,\n" % (i + 1) for i in range(10))
    d2 = ctx.to_device(np.frombuffer(few + bytes(64), np.uint8))
    cols = par.parse_device(d2.ptr, len(few))
    assert cols.n_rows == 10 and cols.n_undecided == 0
    off = np.empty(11, np.int32)
    ctx._check(ctx.lib.exon_hip_memcpy_d2h(ctx.h, off.ctypes.data, cols.list_offsets[0], off.nbytes, None))
    assert off.tolist() == list(range(0, 44, 4))
    bits = np.empty(5, np.uint8)
    ctx._check(ctx.lib.exon_hip_memcpy_d2h(ctx.h, bits.ctypes.data, cols.list_item_valid[0], 5, None))
    assert _bits(bits, 40).tolist() == [False, True, False, False] * 10
    par.close()


@pytest.mark.gpu
def test_gpu_bcf_parser_list_valued_info(ctx, tmp_path):
    """The BCF twin of the test above: typed int8 / int16 / int32 / float vectors -> List<Int32> / List<Float32> on the device
    (missing value -> NULL item, one missing item -> NULL list), next to an exact Int32 scalar."""
    import ctypes as C
    import gzip
    import struct
    import vcf_bcf_writer as W
    from exon_amd import _lib as L
    rows = W.make_rows(20_000, seed=6)
    path = tmp_path / "l.bcf"
    W.write_bcf(path, rows, BGZIP)
    raw = gzip.decompress(open(path, "rb").read())
    l_text, = struct.unpack_from("<I", raw, 5)
    body = raw[9 + l_text:]
    sidx = W.string_index()
    h = C.c_void_p()
    ctx._check(ctx.lib.exon_hip_bcf_parser_create(ctx.h, 2, len(sidx), 0, -1, len(body) + 4096, C.byref(h)))
    keys = (C.c_int32 * 3)(sidx["AC"], sidx["MQS"], sidx["DP"])
    ctx._check(ctx.lib.exon_hip_bcf_parser_set_info_keys(h, keys, b"IFi", 3))
    d = ctx.to_device(np.frombuffer(body + bytes(64), np.uint8))
    cols = L.VCFColumns()
    ctx._check(ctx.lib.exon_hip_bcf_parser_parse(h, None, d.ptr, len(body), C.byref(cols)))
    n = cols.n_rows
    assert n == len(rows) and cols.n_undecided == 0 and cols.n_info == 3 and cols.info_kinds[:3] == b"IFi"

    def dev(ptr, dtype, count):
        out = np.empty(count, dtype)
        if count:
            ctx._check(ctx.lib.exon_hip_memcpy_d2h(ctx.h, out.ctypes.data, ptr, out.nbytes, None))
        return out
    for k, (key, dtype) in enumerate((("AC", np.int32), ("MQS", np.float32))):
        want = W.expected_column(rows, key)
        valid = _bits(dev(cols.infos_valid[k], np.uint8, (n + 7) // 8), n)
        off = dev(cols.list_offsets[k], np.int32, n + 1)
        total = int(off[-1])
        items = dev(cols.infos[k], dtype, total)
        ivalid = _bits(dev(cols.list_item_valid[k], np.uint8, (total + 7) // 8), total) if total else np.zeros(0, bool)
        assert off[0] == 0 and np.all(np.diff(off) >= 0) and total == sum(len(x) for x in want if x is not None)
        got = [None if not valid[r] else [items[i].item() if ivalid[i] else None for i in range(off[r], off[r + 1])] for r in range(n)]
        assert got == want, key
    v = _bits(dev(cols.infos_valid[2], np.uint8, (n + 7) // 8), n)
    vals = dev(cols.infos[2], np.int32, n)
    assert [vals[r].item() if v[r] else None for r in range(n)] == W.expected_column(rows, "DP")
    ctx._check(ctx.lib.exon_hip_bcf_parser_destroy(h))


def _k4(ctx, path, fmt, fields, columns, gpu_parse, thr):
    scan = exon_amd.Scan(str(path), fmt, info_field=fields, gpu_parse=gpu_parse)
    plan = ctx.plan_cmp_avg_by_group(">", thr, 64, columns=columns)
    st = plan.open()
    rows = st.consume(scan)
    counts, sums = st.finish()
    names = scan.dictionary(3)
    res = {names[g]: (int(counts[g]), int(counts[64 + g]), float(sums[g])) for g in range(len(names)) if counts[64 + g]}
    on_gpu = scan.decoded_on_gpu()[0]
    st.close(); plan.close(); scan.close()
    return rows, res, on_gpu


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["vcf", "bcf"])
def test_two_info_fields_through_the_gpu_pipeline(ctx, oracle, fmt):
    """WHERE info.MQ0F > -1 ... AVG(info.DP) GROUP BY filter: x = scan column 5 (second INFO field), y = column 4 (first)."""
    path = os.path.join(FX, fmt, "index." + fmt)
    v = decode.decode_bcf(path) if fmt == "bcf" else decode.decode_vcf(path)
    dp, mq = oracle_typed(v, ["DP", "MQ0F"])
    n = len(dp)
    names = sorted(set(";".join(f) for f in v["filter"]))
    fid = np.array([names.index(";".join(f)) for f in v["filter"]], np.int32)
    pad = np.zeros(64, np.uint8)
    x = np.array([0.0 if a is None else a for a in mq], np.float32)
    y = np.array([0.0 if a is None else a for a in dp], np.float32)
    xv = np.concatenate([np.packbits([a is not None for a in mq], bitorder="little"), pad])
    yv = np.concatenate([np.packbits([a is not None for a in dp], bitorder="little"), pad])
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(x, xv, y, yv, fid, names, -1.0, ">")
    want = {names[g]: (int(cn[g]), int(cr[g]), float(s[g])) for g in range(len(names)) if cr[g]}
    for gpu in (True, False):
        rows, got, on_gpu = _k4(ctx, path, fmt, "DP,MQ0F", (5, 4, 3), gpu, -1.0)
        assert rows == n == 621 and on_gpu == gpu and got.keys() == want.keys()
        for k in want:
            assert got[k][:2] == want[k][:2] and got[k][2] == pytest.approx(want[k][2], rel=1e-9)
        assert sum(v[0] for v in got.values()) == 621 and sum(v[2] for v in got.values()) > 621


def _int_k4_expected(rows, xkey, op, thr, ykey=None):
    """{filter text: (COUNT(y), COUNT(*), SUM(y))} in exact python arithmetic: WHERE info.<xkey> <op> thr GROUP BY filter;
    y = qual, or info.<ykey>"""
    import operator
    cmp = {">": operator.gt, ">=": operator.ge, "<": operator.lt, "<=": operator.le, "=": operator.eq, "!=": operator.ne}[op]
    out = {}
    for r in rows:
        x = None if r["info"] is None else r["info"].get(xkey)
        if x is None or x is True or not cmp(x, thr):
            continue
        y = r["qual"] if ykey is None else (None if r["info"] is None else r["info"].get(ykey))
        k = ";".join(r["filter"])
        cn, cr, sm = out.get(k, (0, 0, 0.0))
        out[k] = (cn + (y is not None), cr + 1, sm + (0.0 if y is None else float(y)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["vcf", "bcf"])
def test_integer_info_predicate_is_exact_beyond_2_pow_24(ctx, tmp_path, fmt):
    """`WHERE info.DP > 16777217` (missing #3): as f32 the literal and the values 16777216..16777219 collapse;
    typed Int32 (schema_builder.rs:197-205) the comparison is exact -- on the GPU decoders and the host decoders, VCF and
    BCF, for every operator, with integer and fractional literals; and AVG(info.DP) sums the integers exactly."""
    import vcf_bcf_writer as W
    rows = W.make_rows(40_000, seed=11)
    path = tmp_path / ("k." + fmt)
    (W.write_vcf(path, rows) if fmt == "vcf" else W.write_bcf(path, rows, BGZIP))
    o = (decode.decode_vcf if fmt == "vcf" else decode.decode_bcf)(str(path))
    assert decode.typed_info(o, "DP")[1] == W.expected_column(rows, "DP")  # the oracle sees the same integers
    cases = [(">", 16777217), (">=", 16777217), ("<", 16777217), ("=", 16777217), ("!=", 16777216), (">", 16777216.5), ("<=", -5),
             (">", 2**31 - 2), (">", 2.0**40), ("<", -2.0**40)]
    for gpu in (True, False):
        for op, thr in cases:
            rows_n, got, on_gpu = _k4_op(ctx, path, fmt, "DP", (4, 2, 3), gpu, op, float(thr))
            want = _int_k4_expected(rows, "DP", op, thr)
            assert rows_n == len(rows) and on_gpu == gpu
            assert {k: v[:2] for k, v in got.items()} == {k: v[:2] for k, v in want.items()}, (fmt, gpu, op, thr)
            for k in want:
                assert got[k][2] == pytest.approx(want[k][2], rel=1e-9)
        # y = info.DP (Int32) under AVG: x = AF (column 4), y = DP (column 5)
        rows_n, got, on_gpu = _k4_op(ctx, path, fmt, "AF,DP", (4, 5, 3), gpu, ">", 0.01)
        want = _int_k4_expected(rows, "AF", ">", float(np.float32(0.01)) if False else 0.01, ykey="DP")
        assert {k: v[:2] for k, v in got.items()} == {k: v[:2] for k, v in want.items()}, (fmt, gpu, "avg(DP)")
        for k in want:
            assert got[k][2] == want[k][2], (fmt, gpu, k)  # integer sums below 2^53: exact in f64


def _k4_op(ctx, path, fmt, fields, columns, gpu_parse, op, thr):
    scan = exon_amd.Scan(str(path), fmt, info_field=fields, gpu_parse=gpu_parse)
    plan = ctx.plan_cmp_avg_by_group(op, thr, 64, columns=columns)
    st = plan.open()
    rows = st.consume(scan)
    counts, sums = st.finish()
    names = scan.dictionary(3)
    res = {names[g]: (int(counts[g]), int(counts[64 + g]), float(sums[g])) for g in range(len(names)) if counts[64 + g]}
    on_gpu = scan.decoded_on_gpu()[0]
    st.close(); plan.close(); scan.close()
    return rows, res, on_gpu


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["vcf", "vcf.gz", "bcf"])
def test_string_and_list_info_keys_the_plan_does_not_read_stay_on_the_gpu(ctx, tmp_path, fmt):
    """A scan that NAMES a String key, a String list and Integer / Float lists (info_field = "AF,CSQ,AC,MQS,TAGS,DP") consumed by a
    plan that reads AF and DP only: the file goes through the GPU pipeline (round 4: any such key sent the whole scan to the host
    decoder) -- the device parser decodes and validates the numeric lists, does not look at the string keys, and the plan's columns
    keep their SCAN positions (DP is scan column 9, behind keys the device skipped).  Expected values straight from the writer's
    rows (tests/vcf_bcf_writer.py: independent of decoders and oracle).  Batches of the same file still come from the host reader
    with every column built."""
    import vcf_bcf_writer as W
    rows = W.make_rows(20000, seed=11)
    path = str(tmp_path / "t.vcf")
    W.write_vcf(path, rows)
    if fmt == "vcf.gz":
        subprocess.check_call([BGZIP, path, path + ".gz"])
        path += ".gz"
    elif fmt == "bcf":
        path = str(tmp_path / "t.bcf")
        W.write_bcf(path, rows, BGZIP)
    kind = "bcf" if fmt == "bcf" else "vcf"
    want = {}
    for r in rows:
        info = r["info"] or {}
        af, dp = info.get("AF"), info.get("DP")
        if af is not None and float(np.float32(af)) > 0.01:
            c = want.setdefault(";".join(r["filter"]), [0, 0, 0.0])
            c[1] += 1
            if dp is not None:
                c[0] += 1
                c[2] += dp
    fields = "AF,CSQ,AC,MQS,TAGS,DP"
    scan = exon_amd.Scan(path, kind, info_field=fields, gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 8, columns=(4, 9, 3))   # info.AF > 0.01, AVG(info.DP) GROUP BY filter
    st = plan.open()
    n = st.consume(scan)
    counts, sums = st.finish()
    names = scan.dictionary(3)
    got = {names[g]: [int(counts[g]), int(counts[8 + g]), float(sums[g])] for g in range(len(names)) if counts[8 + g]}
    assert scan.decoded_on_gpu()[0], "the scan fell back to the host decoder"
    assert n == len(rows) and got.keys() == want.keys()
    for k in want:
        assert got[k][:2] == want[k][:2] and got[k][2] == pytest.approx(want[k][2], rel=1e-12), k
    st.close()
    plan.close()
    scan.close()
    s = exon_amd.Scan(path, kind, info_field=fields, gpu_parse=True)   # batches: the host reader, every column built
    b = list(s)
    assert sum(len(x) for x in b) == len(rows) and sorted(s.dictionary(5)) == ["missense", "stop", "syn"]
    col_ac = [x for bb in b for x in bb.field(6).to_pylist()]
    assert col_ac == W.expected_column(rows, "AC")
    s.close()


@pytest.mark.gpu
def test_a_plan_that_groups_by_a_string_info_key_stays_on_the_gpu(ctx, tmp_path):
    """GROUP BY info.CSQ (a Number=1 String key).  Until round 5 the device did not build that column and the consume went to the host
    decoder; since round 6 the key's dictionary is built on the device like the FILTER one (k_info_string_ids): the consume stays
    on the GPU, the dictionary ids are the group ids, the names come back with the scan."""
    p = tmp_path / "t.vcf"
    p.write_text(HEAD + "".join(f"1\t{i + 1}\t.\tA\tC\t{i}\tPASS\tAF=0.{1 + i % 9};CSQ=c{i % 3}\n" for i in range(5000)))
    scan = exon_amd.Scan(str(p), "vcf", info_field="AF,CSQ", gpu_parse=True)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 8, columns=(4, 2, 5))
    st = plan.open()
    assert st.consume(scan) == 5000 and scan.decoded_on_gpu()[0]
    counts, sums = st.finish()
    names = scan.dictionary(5)
    assert sorted(names) == ["c0", "c1", "c2"] and sorted(int(counts[8 + g]) for g in range(3)) == [1666, 1667, 1667]
    st.close()
    plan.close()
    scan.close()
    s = exon_amd.Scan(str(p), "vcf", info_field="AF,CSQ")
    assert s.dictionary(5) == [] and len(list(s)) == 1 and s.dictionary(5) == ["c0", "c1", "c2"]
    s.close()


@pytest.mark.gpu
def test_string_info_key_dictionary_on_the_device_equals_the_host_readers(ctx, tmp_path, monkeypatch):
    """A String key with missing values, '.', a Character key, and rows whose INFO is '.': batches out of the GPU pipeline carry
    the key as a dictionary column with the host reader's VALUES and NULLs (ids may be numbered differently); a GROUP BY over a key
    with NULLs has a NULL group (the key ""), from the device parser and from the host path alike; more distinct values than
    the device dictionary holds hand the file to the host reader."""
    rng = np.random.default_rng(3)
    head = HEAD.replace("##INFO=<ID=AF", '##INFO=<ID=TYPE,Number=1,Type=Character,Description="t">\n##INFO=<ID=AF')
    lines = []
    for i in range(60000):
        k = i % 11
        info = "." if k == 0 else f"AF=0.{1 + i % 9}" + ("" if k == 1 else f";CSQ={'.' if k == 2 else ('gene' + str(int(rng.integers(0, 40))))}") + (f";TYPE={'SID'[i % 3]}" if k % 2 else "")
        lines.append(f"1\t{i + 1}\t.\tA\tC\t{i % 97}\tPASS\t{info}\n")
    p = tmp_path / "s.vcf"
    p.write_text(head + "".join(lines))
    monkeypatch.setenv("EXON_HIP_GPU_PARSE_SLAB_MB", "1")   # several slabs: the dictionary grows from slab to slab

    def cols(scan):
        out = {}
        for b in scan:
            for i in range(b.type.num_fields):
                out.setdefault(b.type.field(i).name, []).extend(b.field(i).to_pylist())
        return out
    s = exon_amd.Scan(str(p), "vcf", info_field="AF,CSQ,TYPE", gpu_parse=True).bind_ctx(ctx)
    dev = cols(s)
    assert s.decoded_on_gpu()[0]
    assert sorted(s.dictionary(5)) == sorted({v for v in dev["info.CSQ"] if v is not None}) and len(s.dictionary(5)) == 40
    s.close()
    host = cols(exon_amd.Scan(str(p), "vcf", info_field="AF,CSQ,TYPE"))
    for k in ("info.AF", "info.CSQ", "info.TYPE", "pos"):
        assert dev[k] == host[k], k
    assert dev["info.CSQ"].count(None) > 10000 and set(dev["info.TYPE"]) == {"S", "I", "D", None}
    # GROUP BY a key that has NULLs: NULL is a group of its own (DataFusion's GROUP BY), carried as the key "" -- the id of the empty
    # text, which no row can have as a value -- by the device parser and, with EXON_HIP_GPU_PARSE off, by the host path
    want = {}
    for i in range(60000):
        k = i % 11
        if k == 0:
            continue  # INFO '.': AF is NULL, the row does not pass
        key = "" if k in (1, 2) else None
        if key is None:
            key = host["info.CSQ"][i]
        w = want.setdefault(key, [0, 0, 0.0])
        w[0] += 1
        w[1] += 1
        w[2] += float(i % 97)
    assert "" in want and want[""][0] == 2 * (60000 // 11) + (1 if 60000 % 11 > 1 else 0) + (1 if 60000 % 11 > 2 else 0)
    for gpu in (True, False):
        scan = exon_amd.Scan(str(p), "vcf", info_field="AF,CSQ", gpu_parse=gpu)
        plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 5))
        st = plan.open()
        assert st.consume(scan) == 60000 and scan.decoded_on_gpu()[0] == gpu
        counts, sums = st.finish()
        names = scan.dictionary(5)
        got = {names[g]: [int(counts[64 + g]), int(counts[g]), float(sums[g])] for g in range(len(names)) if counts[64 + g]}
        assert got.keys() == want.keys(), gpu
        for k in want:
            assert got[k][:2] == want[k][:2] and got[k][2] == pytest.approx(want[k][2], rel=1e-12), (gpu, k)
        st.close()
        plan.close()
        scan.close()
    # more distinct values than the device dictionary holds: the host reader takes the file (and builds every value)
    q = tmp_path / "many.vcf"
    q.write_text(HEAD + "".join(f"1\t{i + 1}\t.\tA\tC\t1\tPASS\tAF=0.5;CSQ=v{i}\n" for i in range(9000)))
    s = exon_amd.Scan(str(q), "vcf", info_field="AF,CSQ", gpu_parse=True).bind_ctx(ctx)
    many = cols(s)
    assert many["info.CSQ"] == [f"v{i}" for i in range(9000)] and not s.decoded_on_gpu()[0]
    s.close()
