"""CPU: field rules where the first-round host decoders were looser than the reference: a POS that is not
a number is an error (`record.variant_start().transpose()?`, exon-vcf/src/array_builder/lazy_array_builder.rs:163-168), POS 0
is NULL in VCF and BCF alike, and a float field must match Rust's `str::parse::<f32>` grammar as a whole.  The oracle's
decoders (oracle/decode.py) follow the same rules; BCF / SAM oracle decoders are pinned on the reference fixtures."""
import os
import struct
import subprocess

import numpy as np
import pytest

import exon_amd
from oracle import decode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")
HEAD = ('##fileformat=VCFv4.3\n##contig=<ID=1>\n##contig=<ID=2>\n##INFO=<ID=AF,Number=1,Type=Float,Description="x">\n'
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")


def rows_of(path, fmt="vcf", **kw):
    s = exon_amd.Scan(path, fmt, **kw)
    out = [b for b in s]
    s.close()
    return out


def write(tmp_path, body, name="t.vcf"):
    p = tmp_path / name
    p.write_text(HEAD + body)
    return p


@pytest.mark.parametrize("pos", [".", "12x", "", "-5", "1e3", " 7", "+", "++5", "+-5", "5+"])
def test_malformed_pos_is_an_error_like_in_the_reference(tmp_path, pos):
    p = write(tmp_path, f"1\t5\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t{pos}\t.\tA\tC\t1\tPASS\tAF=0.5\n")
    with pytest.raises(exon_amd.ExonHipError, match="POS"):
        rows_of(p)
    with pytest.raises(ValueError):
        decode.decode_vcf(str(p))
    # the pushed-down filter looks at POS only when the contig matches (indexed_async_batch_stream.rs:99-116)
    assert sum(len(b) for b in rows_of(p, region="2:1-100")) == 0
    with pytest.raises(exon_amd.ExonHipError, match="POS"):
        rows_of(p, region="1:1-100")


def test_pos_with_a_leading_plus_is_a_number_as_for_rusts_from_str(tmp_path):
    """`record.variant_start()` (lazy_array_builder.rs:163-168) parses POS with Rust's usize::from_str: noodles-vcf 0.70.0 -- the
    version the reference's Cargo.lock pins (lines 3915-3930) -- depends on no number parser (futures, indexmap, memchr, noodles-*,
    percent-encoding, pin-project-lite, tokio), and core's FromStr for unsigned integers accepts ONE leading '+' ("+5" -> 5, "+0" ->
    0, i.e. NULL here) and nothing else in front of the digits.  Host decoder, device decoder (tests/test_gpu_vcf_parse.py has the
    device twin) and oracle agree."""
    p = write(tmp_path, "1\t+5\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t+0\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t+0012\t.\tA\tC\t1\tPASS\tAF=0.5\n")
    assert rows_of(p)[0].field(1).to_pylist() == [5, None, 12]
    assert decode.decode_vcf(str(p))["pos"] == [5, None, 12]
    assert sum(len(b) for b in rows_of(p, region="1:5-5")) == 1


def test_crlf_lines_lose_their_cr_in_product_and_oracle(tmp_path):
    """A CRLF-terminated VCF: the CR is the line terminator's, not the last INFO value's (product and oracle alike; round 4's
    oracle kept it inside the value)."""
    p = tmp_path / "crlf.vcf"
    p.write_bytes((HEAD + "1\t5\t.\tA\tC\t1.5\tPASS\tAF=0.25\n2\t9\t.\tA\tC\t.\tq10\tAF=0.5\n").replace("\n", "\r\n").encode())
    b = rows_of(p, info_field="AF")[0]
    assert b.field(1).to_pylist() == [5, 9] and [np.float32(x) for x in b.field(4).to_pylist()] == [np.float32(0.25), np.float32(0.5)]
    v = decode.decode_vcf(str(p))
    assert v["pos"] == [5, 9] and [i["AF"] for i in v["info"]] == ["0.25", "0.5"] and v["filter"] == [["PASS"], ["q10"]]


def test_pos_zero_is_null_in_vcf_and_oracle(tmp_path):
    p = write(tmp_path, "1\t0\t.\tA\tC\t1\tPASS\tAF=0.5\n1\t7\t.\tA\tC\t.\t.\t.\n")
    b = rows_of(p)[0]
    assert b.field(1).to_pylist() == [None, 7]
    assert decode.decode_vcf(str(p))["pos"] == [None, 7]


@pytest.mark.parametrize("text,ok", [("1.5", True), ("1e-3", True), ("inf", True), ("-Infinity", True), ("NaN", True), (".5", True),
                                      ("5.", True), ("1.5abc", False), ("0x1p3", False), (" 1.5", False), ("1.5 ", False),
                                      ("1e", False), ("--1", False), ("0." + "3" * 80, True), ("1" + "0" * 70 + "x", False)])
def test_float_fields_follow_rusts_grammar(tmp_path, text, ok):
    p = write(tmp_path, f"1\t5\t.\tA\tC\t{text}\tPASS\tAF={text}\n")
    if ok:
        b = rows_of(p, info_field="AF")[0]
        want = np.float32(float(text))
        got_q, got_i = np.float32(b.field(2).to_pylist()[0]), np.float32(b.field(4).to_pylist()[0])
        assert (np.isnan(want) and np.isnan(got_q) and np.isnan(got_i)) or (got_q == want and got_i == want)
    else:
        with pytest.raises(exon_amd.ExonHipError, match="float"):
            rows_of(p, info_field="AF")


def test_bcf_pos0_minus_one_is_null(tmp_path):
    """A BCF record with pos0 = -1 (VCF POS 0): NULL in the host decoder, like the VCF path, and in the oracle."""
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    ub, bcf = tmp_path / "s.ubcf", tmp_path / "s.bcf"
    subprocess.check_call([gen, "bcf", "50", str(ub)])
    raw = bytearray(open(ub, "rb").read())
    l_text = struct.unpack_from("<I", raw, 5)[0]
    o = 9 + l_text
    for _ in range(3):  # the 4th record
        l_shared, l_indiv = struct.unpack_from("<II", raw, o)
        o += 8 + l_shared + l_indiv
    struct.pack_into("<i", raw, o + 8 + 4, -1)
    open(ub, "wb").write(bytes(raw))
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "bgzip"), str(ub), str(bcf), "6"])
    pos = [x for b in rows_of(bcf, "bcf") for x in b.field(1).to_pylist()]
    want = decode.decode_bcf(str(bcf))["pos"]
    assert pos == want and pos[3] is None and pos[2] is not None and len(pos) == 50


def test_oracle_bcf_decoder_is_pinned_on_the_reference_fixture():
    """index.bcf is the BCF twin of index.vcf (exon-core/src/session_context/exon_context_ext.rs:1053-1090: 621 records, 191 in
    region '1'): the oracle's BCF decoder must give the columns its VCF decoder gives for the twin."""
    b, v = decode.decode_bcf(os.path.join(FX, "bcf", "index.bcf")), decode.decode_vcf(os.path.join(FX, "vcf", "index.vcf"))
    assert len(b["chrom"]) == 621 and sum(c == "1" for c in b["chrom"]) == 191
    assert b["chrom"] == v["chrom"] and b["pos"] == v["pos"] and b["filter"] == v["filter"]
    assert [None if q is None else np.float32(q) for q in b["qual"]] == [None if q is None else np.float32(q) for q in v["qual"]]
    for bi, vi in zip(b["info"], v["info"]):
        assert np.float32(bi["MQ0F"]) == np.float32(vi["MQ0F"]) and int(bi["DP"]) == int(vi["DP"])


def test_oracle_sam_decoder_on_the_reference_fixture():
    refs, recs = decode.decode_sam(os.path.join(FX, "sam", "test.sam"))
    assert refs == [("ref1", 56)] and len(recs) == 1
    r = recs[0]
    assert (r["name"], r["flag"], r["ref_id"], r["start"], r["end"], r["mapq"], r["cigar"]) == ("ref1_grp1_p001", 99, 0, 1, 10, 0, "10M")
    # and the product's host SAM reader agrees
    b = rows_of(os.path.join(FX, "sam", "test.sam"), "sam")[0]
    assert [b.field(i).to_pylist()[0] for i in range(5)] == [99, 0, "ref1", 1, 10]
