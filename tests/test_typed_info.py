"""Typed INFO fields (InfosBuilder, exon-vcf/src/array_builder/info_builder.rs:152-309 + typing in
exon-core/src/datasources/vcf/schema_builder.rs:197-249): several keys per scan, Float / Integer -> f32, Flag -> Boolean (true
when present, NULL when absent), String / Character -> dictionary, whole struct NULL when INFO is '.'.  CPU tests compare
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
    """per field: python values (float / True / str / None) from the oracle's decoded INFO dicts + header types"""
    out = []
    for f in fields:
        number, typ = v["info_header"][f]
        col = []
        for i in v["info"]:
            x = None if i is None else i.get(f)
            if typ == "Flag":
                col.append(True if x is not None else None)
            elif x is None or x is True or x == ".":
                col.append(None)
            elif typ in ("Float", "Integer"):
                col.append(float(np.float32(x)))
            else:
                col.append(x)
        out.append(col)
    return out


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
    assert types == ["float", "float", "bool", "dictionary<values=string, indices=int32, ordered=0>"]
    want = oracle_typed(decode.decode_vcf(str(p)), fields)
    assert got == want
    assert got[0] == [0.5, None, None, 0.25, pytest.approx(1e-3), None, None]        # first occurrence wins; '.' -> NULL
    assert got[2] == [True, None, None, True, None, True, None]                        # Flag: true when present, NULL when absent
    assert got[3] == ["missense", None, None, None, "stop", None, "missense"]
    assert host_typed(p, "vcf", ["AA"])[0] == [["T", None, None, None, None, None, None]]
    with pytest.raises(exon_amd.ExonHipError, match="AC"):                              # list-valued fields are not scalars
        host_typed(p, "vcf", ["AC"])
    with pytest.raises(exon_amd.ExonHipError, match="NOPE"):
        host_typed(p, "vcf", ["NOPE"])
    with pytest.raises(exon_amd.ExonHipError, match="at most 4"):
        host_typed(p, "vcf", ["AF", "DP", "DB", "CSQ", "AA"])


def test_reference_fixture_typed_info_vcf_and_bcf_twin():
    fields = ["DP", "MQ0F", "INDEL", "IDV"]
    v = decode.decode_vcf(os.path.join(FX, "vcf", "index.vcf"))
    want = oracle_typed(v, fields)
    got, types = host_typed(os.path.join(FX, "vcf", "index.vcf"), "vcf", fields)
    assert types == ["float", "float", "bool", "float"] and got == want
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
    par = exon_amd.VCFParser(ctx, ["1"], info_field="AF,DP:f,DB:b")
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
            vals = dev(cols.infos[k], np.float32, n)
            assert valid.tolist() == [x is not None for x in want[k]]
            assert [float(v) for v, ok in zip(vals, valid) if ok] == [x for x in want[k] if x is not None]
    par.close()


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


@pytest.mark.gpu
def test_string_info_field_keeps_the_scan_on_the_host(ctx, tmp_path):
    p = tmp_path / "t.vcf"
    p.write_text(HEAD + "".join(f"1\t{i + 1}\t.\tA\tC\t{i}\tPASS\tAF=0.{1 + i % 9};CSQ=c{i % 3}\n" for i in range(5000)))
    rows, got, on_gpu = _k4(ctx, p, "vcf", "AF,CSQ", (4, 2, 3), True, 0.01)
    assert rows == 5000 and not on_gpu and got["PASS"][1] == 5000
    s = exon_amd.Scan(str(p), "vcf", info_field="AF,CSQ")
    assert s.dictionary(5) == [] and len(list(s)) == 1 and s.dictionary(5) == ["c0", "c1", "c2"]
    s.close()
