"""Pins the CPU oracle on every value the reference's own tests hold for the scan->filter->aggregate path
(SURVEY.md section 8c).  Runs on CPU.  Each test cites the reference test it restates
(paths relative to /root/reference/exon/exon-core)."""
import os

import numpy as np
import pytest

from oracle import decode

FX = os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures")


def fx(*p):
    return os.path.join(FX, *p)


# ---- UDF truth tables: tests/sqllogictests/slt/vcf-udfs.slt:1-32 ----------------------------------------
INTERVALS = [("1", 1), ("1", 1), ("1", 2), ("2", 2), ("2", 3)]


def test_region_match_truth_table(oracle):
    assert [oracle.region_match(c, p, "1:1-1") for c, p in INTERVALS] == [True, True, False, False, False]


def test_interval_match_truth_table(oracle):
    assert [oracle.interval_match(p, "1-1") for _, p in INTERVALS] == [True, True, False, False, False]
    assert oracle.interval_match(None, "1-1") is False  # udfs/vcf/mod.rs:267: NULL -> false


def test_chrom_match_truth_table(oracle):
    assert [oracle.chrom_match(c, "1") for c, _ in INTERVALS] == [True, True, True, False, False]


def test_region_match_null_is_an_error(oracle):
    with pytest.raises(ValueError):  # udfs/vcf/mod.rs:107-110
        oracle.region_match(None, 1, "1:1-1")
    with pytest.raises(ValueError):
        oracle.region_match("1", None, "1:1-1")


# ---- physical-expr KATs ---------------------------------------------------------------------------------
def test_region_physical_expr_evaluate(oracle):
    """src/physical_plan/region_physical_expr.rs:306-345: chr1:1-1 over [(chr1,1),(chr1,2),(chr2,3)] -> [T,F,F]."""
    names = ["chr1", "chr2"]
    chrom = np.array([0, 0, 1], np.int32)
    pos = np.array([1, 2, 3], np.int64)
    got = [oracle.c2_region_count(chrom[i:i + 1], pos[i:i + 1], names, "chr1:1-1")[0] for i in range(3)]
    assert got == [1, 0, 0]


def test_pos_interval_physical_expr_evaluate(oracle):
    """src/physical_plan/pos_interval_physical_expr.rs:277-316: pos = 1 over [1,2,3] -> [T,F,F]."""
    assert [oracle.interval_match(p, "1-1") for p in (1, 2, 3)] == [True, False, False]


def test_start_end_interval_expr_construction(oracle):
    """src/physical_plan/start_end_interval_physical_expr.rs:222-268 (test_from_start_expr / test_from_end_expr)."""
    assert oracle.start_end_interval_from_expr("start", ">", 4) == (4, None)
    assert oracle.start_end_interval_from_expr("end", "<", 4) == (0, 4)
    for col, op in (("start", "<"), ("end", ">"), ("pos", ">"), ("start", ">=")):
        with pytest.raises(ValueError):
            oracle.start_end_interval_from_expr(col, op, 4)


def test_region_grammar(oracle):
    """datasources/vcf/table_provider.rs:583-587 region strings; region_physical_expr.rs:91-104 defaults."""
    assert oracle.parse_region("1") == ("1", 1, None)
    assert oracle.parse_region("chr1:1-12209145") == ("chr1", 1, 12209145)
    assert oracle.parse_region("1:9999921") == ("1", 9999921, None)
    assert oracle.parse_region("chr1:10000-10000000") == ("chr1", 10000, 10000000)
    assert oracle.parse_region("HLA-A*01:01") == ("HLA-A*01", 1, None)  # ':01' parses as start = 1
    assert oracle.parse_region("a:b") == ("a:b", 1, None)               # invalid suffix -> whole string is the name


# ---- quality scores: tests/sqllogictests/slt/quality-score-udfs.slt:1-23 ---------------------------------
def test_quality_scores_to_list(oracle):
    assert oracle.quality_scores_to_list("###") == [2, 2, 2]
    assert oracle.quality_scores_to_list("!\"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHI") == list(range(41))


# ---- SAM flags: src/udfs/sam/samflags.rs:111-135 ---------------------------------------------------------
def test_sam_flag_bits(oracle):
    bits = dict(segmented=0x1, properly=0x2, unmapped=0x4, mate_unmapped=0x8, reverse=0x10, mate_reverse=0x20,
                first=0x40, last=0x80, secondary=0x100, qc_fail=0x200, duplicate=0x400, supplementary=0x800)
    flag = 83  # first BAM row (slt/bam-select-tests.slt:12): paired, proper, reverse, first
    want = dict.fromkeys(bits, False)
    want.update(segmented=True, properly=True, reverse=True, first=True)
    assert {k: oracle.sam_flag(flag, b) for k, b in bits.items()} == want
    assert oracle.sam_flag(0x10000 | 4, 4) and not oracle.sam_flag(0x10000, 0xFFFF)  # `flag as u16` truncation


# ---- VCF: slt/vcf-select-tests.slt:47-55, slt/vcf-indexed-tests.slt:22-59 --------------------------------
@pytest.mark.parametrize("name", ["index.vcf", "index.vcf.gz"])
def test_vcf_count_621(oracle, name):
    v = decode.decode_vcf(fx("vcf", name))
    assert len(v["chrom"]) == 621
    names, chrom_id, pos, pv = decode.vcf_device_columns(v)
    per = {n: oracle.c2_region_count(chrom_id, pos, names, n, pos_valid=pv)[0] for n in ("1", "2", "10", "a")}
    assert per == {"1": 191, "2": 219, "10": 211, "a": 0}  # region '1' -> 191 (vcf-indexed-tests.slt:31-35)


def test_vcf_partition_region_counts(oracle):
    total = 0
    for sample, f in (("1", "index1.vcf.gz"), ("2", "index2.vcf.gz")):
        v = decode.decode_vcf(fx("vcf-partition", f"sample={sample}", f))
        names, chrom_id, pos, pv = decode.vcf_device_columns(v)
        c, _ = oracle.c2_region_count(chrom_id, pos, names, "1", pos_valid=pv)
        assert c == 191  # slt/vcf-indexed-tests.slt:31-35 (sample = '1')
        total += c
        assert oracle.c2_region_count(chrom_id, pos, names, "a", pos_valid=pv)[0] == 0  # :24-28
    assert total == 382  # slt/vcf-indexed-tests.slt:37-40


def test_biobear_vcf_region(oracle):
    v = decode.decode_vcf(fx("biobear-vcf", "vcf_file.vcf.gz"))
    names, chrom_id, pos, pv = decode.vcf_device_columns(v)
    assert oracle.c2_region_count(chrom_id, pos, names, "1", pos_valid=pv)[0] == 11      # :56-59
    assert oracle.c2_region_count(chrom_id, pos, names, "1000", pos_valid=pv)[0] == 0    # :51-54


def test_vcf_first_rows(oracle):
    """slt/vcf-select-tests.slt:6-10: info of the first two records."""
    v = decode.decode_vcf(fx("vcf", "index.vcf"))
    assert v["info"][0]["DP"] == "1" and v["info"][0]["MQ0F"] == "0"
    assert v["info"][0]["I16"] == "1,0,0,0,26,676,0,0,60,3600,0,0,0,0,0,0"
    assert v["info"][1]["I16"].startswith("1,0,0,0,34,1156")


# ---- BAM: slt/bam-select-tests.slt:9-64, slt/bam-indexed-select-tests.slt:11-50 --------------------------
def test_bam_count_and_first_row(oracle):
    refs, recs = decode.decode_bam(fx("bam", "test.bam"))
    assert len(recs) == 61
    r = recs[0]
    rn = [n for n, _ in refs]
    assert (r["name"], r["flag"], rn[r["ref_id"]], r["start"], r["end"], r["mapq"], r["cigar"], rn[r["mate_ref_id"]]) == \
        ("READ_ID", 83, "chr1", 12203704, 12217173, None, "55M13394N21M", "chr1")
    assert r["sequence"] == "A" * 76
    assert [x["quality_score"][0] for x in recs[:5]] == [23, 20, 37, 34, 31]
    assert [len(x["quality_score"]) for x in recs[:5]] == [76] * 5


def test_bam_region_hits(oracle):
    total = 0
    for f in (fx("bam", "test.bam"), fx("bam-multifile", "test2.bam")):
        refs, recs = decode.decode_bam(f)
        rid = [n for n, _ in refs].index("chr1")
        hits = sum(oracle.bam_intersects(r["ref_id"], r["start"], r["end"], rid, 1, 12209145) for r in recs)
        assert hits == 7  # slt/bam-indexed-select-tests.slt:16-19
        total += hits
    assert total == 14    # :26-29 (two files), :47-50


def test_c6_overlap_count_matches_per_record_rule(oracle):
    """The column-wise restatement used against K6 equals the per-record orc_bam_intersects on the fixture (7 hits,
    slt/bam-indexed-select-tests.slt:16-19) and on random intervals with NULLs."""
    refs, recs = decode.decode_bam(fx("bam", "test.bam"))
    names = [n for n, _ in refs]

    def cols(recs):
        n = len(recs)
        rid = np.array([r["ref_id"] if r["ref_id"] is not None else -1 for r in recs], np.int32)
        st = np.array([r["start"] or 0 for r in recs], np.int64)
        en = np.array([r["end"] or 0 for r in recs], np.int64)
        pack = lambda v: np.packbits(np.array(v, bool), bitorder="little")  # noqa: E731
        return (rid, pack([r["ref_id"] is not None for r in recs]), st, pack([r["start"] is not None for r in recs]), en,
                pack([r["end"] is not None for r in recs]), n)

    rid, rv, st, sv, en, ev, n = cols(recs)
    assert oracle.c6_overlap_count(rid, rv, st, sv, en, ev, names, "chr1:1-12209145") == 7
    rng = np.random.default_rng(6)
    fake = [dict(ref_id=(int(rng.integers(0, 3)) if rng.random() > 0.1 else None),
                 start=(int(s0) if rng.random() > 0.1 else None), end=(int(s0 + rng.integers(0, 200)) if rng.random() > 0.1 else None))
            for s0 in rng.integers(1, 5000, 3000)]
    rid, rv, st, sv, en, ev, n = cols(fake)
    for region, (k, a, b) in {"chr2:1000-2000": (1, 1000, 2000), "chr1": (0, 1, None), "chr3:4000": (2, 4000, None)}.items():
        want = sum(oracle.bam_intersects(r["ref_id"], r["start"], r["end"], k, a, b) for r in fake)
        assert oracle.c6_overlap_count(rid, rv, st, sv, en, ev, ["chr1", "chr2", "chr3"], region) == want


def test_bam_group_count_sums_to_file_rows(oracle):
    """No reference test pins GROUP BY reference; the identity sum(groups) == COUNT(*) = 61 must hold."""
    refs, recs = decode.decode_bam(fx("bam", "test.bam"))
    n, flag, mapq, mv, ref, rv = decode.bam_device_columns(recs)
    # predicate that keeps everything with a non-NULL mapq; every read of the fixture has mapq 255 -> NULL -> 0 rows
    cnt, _ = oracle.c3_flag_mapq_group_count(flag, mapq, mv, ref, rv, [x for x, _ in refs], 0, 0, 0)
    assert cnt.sum() == sum(r["mapq"] is not None for r in recs)


# ---- FASTQ / FASTA: slt/fastq-scan-test.slt:6-10,51-83, slt/fasta-scan-tests.slt:6-10,31-34,72-80 ---------
@pytest.mark.parametrize("name", ["test.fastq", "test.fastq.gz", "test_bgzip.fastq.gz"])
def test_fastq_fixture(oracle, name):
    recs = decode.decode_fastq(fx("fastq", name))
    assert len(recs) == 2
    assert (recs[0]["name"], recs[0]["description"]) == ("SEQ_ID", "This is a description")
    assert (recs[1]["name"], recs[1]["description"]) == ("SEQ_ID2", None)
    q = "!''*((((***+))%%%++)(%%%%).1***-+*''))**55CCF>>>>>>CCCCCCC65"
    assert recs[0]["quality_scores"] == q and recs[1]["quality_scores"] == q
    off, data = decode.fastq_device_columns(recs)
    h, _ = oracle.c5_qual_pos_hist(off, data, len(q))
    for p, ch in enumerate(q):  # both reads carry the same string: every position has one bin with count 2
        assert h[p, ord(ch)] == 2 and h[p].sum() == 2
    assert oracle.quality_scores_to_list(q)[:4] == [0, 6, 6, 9]


@pytest.mark.parametrize("name", ["test.fasta", "test.fasta.gz"])
def test_fasta_fixture(name):
    recs = decode.decode_fasta(fx("fasta", name))
    assert [(r["id"], r["description"], r["sequence"]) for r in recs] == [("a", "description", "ATCG"),
                                                                         ("b", "description2", "ATCG")]


# ---- repartition rule: src/datasources/exon_file_scan_config.rs:79-110 -------------------------------------
def test_regroup_files_by_size(oracle):
    assert oracle.regroup_files_by_size([50, 10, 30, 20, 40], 2) == [[1, 2, 0], [3, 4]]
    assert oracle.regroup_files_by_size([5, 5, 5], 8) == [[0], [1], [2]]  # min(target, n_files) groups
    assert oracle.regroup_files_by_size([], 4) == []


# ---- float / NULL semantics cross-checked with pyarrow compute (independent Arrow implementation) ---------
def test_c4_against_pyarrow(oracle):
    import pyarrow as pa
    import pyarrow.compute as pc
    n = 200_000
    af, av, q, qv, fid = oracle.gen_c4(4, 0, n)
    filters = oracle.c4_filters()
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(af, av, q, qv, fid, filters, 0.01, ">")
    avb = np.unpackbits(av, bitorder="little")[:n].astype(bool)
    qvb = np.unpackbits(qv, bitorder="little")[:n].astype(bool)
    t = pa.table({"af": pa.array(af, mask=~avb), "qual": pa.array(q, mask=~qvb),
                  "filter": pa.array([filters[i] for i in fid])})
    keep = pc.greater(pc.cast(t["af"], pa.float64()), pa.scalar(0.01, pa.float64()))
    g = t.filter(keep).group_by("filter").aggregate([("qual", "mean"), ("qual", "count"), ([], "count_all")])
    got = {r["filter"]: r for r in g.to_pylist()}
    for i, name in enumerate(filters):
        assert got[name]["count_all"] == cr[i] and got[name]["qual_count"] == cn[i]
        assert got[name]["qual_mean"] == pytest.approx(s[i] / cn[i], rel=1e-12)
    # the f32-vs-f64 coercion trap: f32(0.01) widened is below the literal
    assert not pc.greater(pc.cast(pa.array([np.float32(0.01)]), pa.float64()), pa.scalar(0.01)).to_pylist()[0]


def _total_order_key_f64(x):
    """IEEE 754 totalOrder as an integer sort key, computed with integer bit tricks only (no float compare): flip all bits
    of negatives, only the sign bit of positives.  -NaN < -inf < ... < -0 < +0 < ... < +inf < +NaN."""
    b = np.asarray(x, np.float64).view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    return np.where(neg, ~b, b | np.uint64(1 << 63))


@pytest.mark.parametrize("op", [">", ">=", "<", "<=", "=", "!="])
@pytest.mark.parametrize("thr", [0.0, -0.0, 0.01, float("inf"), float("nan")])
def test_c4_float_compare_is_total_order_checked_without_the_oracle(oracle, op, thr):
    """arrow-rs' `cmp` kernels (what DataFusion 44's BinaryExpr runs for `CAST(af AS DOUBLE) <op> lit`) order floats by IEEE
    totalOrder: NaN above +inf, -0 below +0, NaN = NaN.  pyarrow's compare is plain IEEE (NaN unordered), so it cannot check
    this; the expectation here comes from integer sort keys built with numpy bit operations -- an implementation that shares
    nothing with oracle/exon_oracle.c's f64 total_cmp."""
    rng = np.random.default_rng(17)
    n = 60_000
    specials = np.array([np.nan, -np.nan, 0.0, -0.0, np.inf, -np.inf, 0.01, 1e-45, -1e-45, 3.4e38, 0.25, -7.5], np.float32)
    af = specials[rng.integers(0, len(specials), n)]
    q = (rng.integers(0, 1000, n) / 8).astype(np.float32)       # eighths: every partial sum is exact in f64
    fid = rng.integers(0, 3, n).astype(np.int32)
    avb, qvb = rng.random(n) < 0.9, rng.random(n) < 0.8
    av = np.concatenate([np.packbits(avb, bitorder="little"), np.zeros(64, np.uint8)])
    qv = np.concatenate([np.packbits(qvb, bitorder="little"), np.zeros(64, np.uint8)])
    names = ["a", "b;c", ""]
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(af, av, q, qv, fid, names, thr, op)
    kx, kt = _total_order_key_f64(af.astype(np.float64)), _total_order_key_f64(np.float64(thr))
    keep = {">": kx > kt, ">=": kx >= kt, "<": kx < kt, "<=": kx <= kt, "=": kx == kt, "!=": kx != kt}[op] & avb
    for g in range(3):
        m = keep & (fid == g)
        assert cr[g] == int(m.sum()) and cn[g] == int((m & qvb).sum())
        assert s[g] == float(q[m & qvb].astype(np.float64).sum())  # exact: eighths below 2^53


def test_c3_against_numpy(oracle):
    n = 300_000
    f, mq, mv, ref, rv = oracle.gen_c3(3, 0, n)
    refs = oracle.c3_refs()
    mvb = np.unpackbits(mv, bitorder="little")[:n].astype(bool)
    rvb = np.unpackbits(rv, bitorder="little")[:n].astype(bool)
    for mask, value, qmin in [(1284, 0, 30), (4, 4, 0), (0, 0, 0)]:
        cnt, _ = oracle.c3_flag_mapq_group_count(f, mq, mv, ref, rv, refs, mask, value, qmin)
        keep = ((f & mask) == value) & mvb & (mq.astype(np.int32) >= qmin)
        want = np.bincount(np.where(rvb, ref, len(refs))[keep], minlength=len(refs) + 1)
        assert np.array_equal(cnt, want)
