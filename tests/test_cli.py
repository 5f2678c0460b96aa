"""exon-hip-cli.  CPU part: config-1 plumbing -- COUNT(*) over FASTA/FASTQ/VCF/BAM via
table functions and external tables, pinned by the reference's slt files.  GPU part: the fused query shapes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "exon_amd", "bin", "exon-hip-cli")
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")


def run(*cmds, ok=True):
    r = subprocess.run([CLI, "-q", "-c", *cmds], capture_output=True, text=True, timeout=600)
    if ok:
        assert r.returncode == 0, r.stderr
    return r


def cells(out):
    rows = [[c.strip() for c in line.strip("|").split("|")] for line in out.splitlines() if line.startswith("|")]
    return rows[1:]  # drop the header of the (single) table


def last_count(out):
    return int(cells(out)[-1][0])


# ---- CPU: slt-pinned counts -----------------------------------------------------------------------------
@pytest.mark.parametrize("sql,want", [
    (f"SELECT COUNT(*) FROM fasta_scan('{FX}/fasta/test.fasta')", 2),                # slt/fasta-scan-tests.slt:72-75
    (f"SELECT COUNT(*) FROM fasta_scan('{FX}/fasta/test.fasta.gz', 'gzip')", 2),     # :82-85
    (f"SELECT COUNT(*) FROM fastq_scan('{FX}/fastq/test.fastq')", 2),                # slt/fastq-scan-test.slt:51-54
    (f"SELECT COUNT(*) FROM fastq_scan('{FX}/fastq/test_bgzip.fastq.gz', 'gzip')", 2),  # :66-69
    (f"SELECT COUNT(*) FROM vcf_scan('{FX}/vcf/index.vcf')", 621),                   # slt/vcf-select-tests.slt:47-50
    (f"SELECT COUNT(*) FROM vcf_scan('{FX}/vcf/index.vcf.gz', 'gzip')", 621),        # :52-55
    (f"SELECT COUNT(*) FROM bam_scan('{FX}/bam/test.bam')", 61),                     # slt/bam-select-tests.slt:56-59
    (f"SELECT COUNT(*) FROM bam_scan('{FX}/bam-multifile')", 122),                   # :61-64 (two files)
    (f"SELECT COUNT(*) FROM vcf_indexed_scan('{FX}/vcf-partition', '1')", 382),      # slt/vcf-indexed-tests.slt:42-45
    (f"SELECT COUNT(*) FROM bam_indexed_scan('{FX}/bam-multifile/', 'chr1:1-12209145')", 14),  # slt/bam-indexed-select-tests.slt:52-55
])
def test_count_star_table_functions(sql, want):
    assert last_count(run(sql).stdout) == want


def test_external_tables_and_region_pushdown():
    out = run(f"CREATE EXTERNAL TABLE fasta_table STORED AS FASTA LOCATION '{FX}/fasta/test.fasta.gz' OPTIONS (compression 'gzip');"
              "SELECT COUNT(*) FROM fasta_table; DROP TABLE fasta_table;").stdout
    assert last_count(out) == 2  # slt/fasta-scan-tests.slt:20-34
    t = f"CREATE EXTERNAL TABLE vcf_table STORED AS INDEXED_VCF LOCATION '{FX}/vcf-partition' OPTIONS (compression gzip);"
    assert last_count(run(t + "SELECT COUNT(*) AS cnt FROM vcf_table WHERE vcf_region_filter('1', chrom) = true").stdout) == 382
    assert last_count(run(t + "SELECT COUNT(*) FROM vcf_table WHERE vcf_region_filter('a', chrom) = true").stdout) == 0
    b = f"CREATE EXTERNAL TABLE bam STORED AS INDEXED_BAM LOCATION '{FX}/bam/';"
    assert last_count(run(b + "SELECT COUNT(*) AS cnt FROM bam WHERE bam_region_filter('chr1:1-12209145', reference, start, end) = true").stdout) == 7
    bb = f"CREATE EXTERNAL TABLE v STORED AS INDEXED_VCF LOCATION '{FX}/biobear-vcf/vcf_file.vcf.gz' OPTIONS (compression gzip);"
    assert last_count(run(bb + "SELECT COUNT(*) FROM v WHERE vcf_region_filter('1', chrom) = true").stdout) == 11
    assert last_count(run(bb + "SELECT COUNT(*) FROM v WHERE vcf_region_filter('1000', chrom) = true").stdout) == 0


def test_cram_external_table_counts_its_records():
    """STORED AS CRAM (exon-core/src/datasources/cram/table_provider.rs): the host CRAM 3.0 decoder behind the CLI; 15 / 910 are
    the sums of the container headers' record counts, and the rows the oracle's decoder returns (tests/test_cram.py)"""
    for name, want in (("test_input_1_a.cram", 15), ("1404_index_multislice.cram", 910)):
        out = run(f"CREATE EXTERNAL TABLE c STORED AS CRAM LOCATION '{FX}/cram/{name}'; SELECT COUNT(*) FROM c;").stdout
        assert last_count(out) == want


@pytest.mark.gpu
def test_cli_bam_region_filter_on_a_plain_table_runs_k6():
    """A plain (not INDEXED_) BAM table: the interval predicate runs on the GPU (K6) over the GPU-decoded columns."""
    b = f"CREATE EXTERNAL TABLE bam STORED AS BAM LOCATION '{FX}/bam/test.bam';"
    q = "SELECT COUNT(*) FROM bam WHERE bam_region_filter('chr1:1-12209145', reference, start, end) = true"
    assert last_count(run(b + q).stdout) == 7
    assert last_count(run(b + "SELECT COUNT(*) FROM bam WHERE bam_region_filter('chr1', reference, start, end) = true").stdout) == 61


def test_indexed_table_without_region_is_an_error():
    """slt/vcf-indexed-tests.slt:6-8 and :48-49 (`statement error`)."""
    r = run(f"CREATE EXTERNAL TABLE v STORED AS INDEXED_VCF LOCATION '{FX}/vcf/index.vcf.gz' OPTIONS (compression gzip); SELECT COUNT(*) FROM v", ok=False)
    assert r.returncode != 0 and "region" in r.stderr


def test_config1_thousand_record_fasta(tmp_path):
    """BASELINE.json configs[0]: SELECT COUNT(*) over a 1k-record FASTA via the CLI on CPU (no GPU involved)."""
    p = tmp_path / "syn.fasta"
    rng = np.random.default_rng(1)
    with open(p, "w") as f:
        for i in range(1000):
            seq = "".join(rng.choice(list("ACGT"), 150))
            f.write(f">rec{i} synthetic\n" + "\n".join(seq[j:j + 60] for j in range(0, 150, 60)) + "\n")
    assert last_count(run(f"SELECT COUNT(*) FROM fasta_scan('{p}')").stdout) == 1000
    assert last_count(run(f"CREATE EXTERNAL TABLE t STORED AS FASTA LOCATION '{p}'; SELECT COUNT(*) FROM t").stdout) == 1000


def test_unsupported_shapes_fail_loudly():
    r = run(f"SELECT id FROM fasta_scan('{FX}/fasta/test.fasta')", ok=False)
    assert r.returncode != 0 and "not supported" in r.stderr
    assert run("SELECT COUNT(*) FROM nope", ok=False).returncode != 0


# ---- GPU: the fused shapes end to end (text file -> native decoder -> HBM -> kernel) ------------------------
@pytest.mark.gpu
def test_cli_region_count_on_fixture(oracle):
    from oracle import decode
    v = decode.decode_vcf(f"{FX}/vcf/index.vcf")
    out = run(f"SELECT COUNT(*) FROM vcf_scan('{FX}/vcf/index.vcf') WHERE chrom = '1' AND pos >= 9999921 AND pos <= 10000000").stdout
    want = sum(1 for c, p in zip(v["chrom"], v["pos"]) if c == "1" and 9999921 <= p <= 10000000)
    assert last_count(out) == want and want > 0
    out = run(f"SELECT COUNT(*) FROM vcf_scan('{FX}/vcf/index.vcf') WHERE region_match(chrom, pos, '2')").stdout
    assert last_count(out) == 219


@pytest.mark.gpu
def test_cli_bam_group_by_reference(oracle):
    from oracle import decode
    refs, recs = decode.decode_bam(f"{FX}/bam/test.bam")
    out = run(f"SELECT reference, COUNT(*) FROM bam_scan('{FX}/bam/test.bam') WHERE flag & 4 = 0 AND CAST(mapping_quality AS INT) >= 0 GROUP BY reference").stdout
    n, flag, mapq, mv, ref, rv = decode.bam_device_columns(recs)
    cnt, _ = oracle.c3_flag_mapq_group_count(flag, mapq, mv, ref, rv, [x for x, _ in refs], 4, 0, 0)
    got = {r[0]: int(r[1]) for r in cells(out)}
    want = {refs[i][0]: int(c) for i, c in enumerate(cnt[:-1]) if c}
    assert got == want


@pytest.mark.gpu
def test_cli_config4_on_synthetic_vcf_text(tmp_path, oracle):
    n = 50_000
    af, av, q, qv, fid = oracle.gen_c4(4, 0, n)
    avb = np.unpackbits(av, bitorder="little")[:n].astype(bool)
    qvb = np.unpackbits(qv, bitorder="little")[:n].astype(bool)
    filters = oracle.c4_filters()
    p = tmp_path / "syn.vcf"
    with open(p, "w") as f:
        f.write('##fileformat=VCFv4.3\n##contig=<ID=1>\n##INFO=<ID=AF,Number=1,Type=Float,Description="AF">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for i in range(n):
            info = f"AF={np.format_float_scientific(af[i], unique=True)}" if avb[i] else "."
            f.write(f"1\t{i + 1}\t.\tA\tC\t{repr(float(q[i])) if qvb[i] else '.'}\t{filters[fid[i]] or '.'}\t{info}\n")
    out = run("SET exon.vcf_parse_info = true;"
              f"CREATE EXTERNAL TABLE v STORED AS VCF LOCATION '{p}';"
              'SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" > 0.01 GROUP BY filter').stdout
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(af, av, q, qv, fid, filters, 0.01, ">")
    got = {r[0]: (float(r[1]), int(r[2])) for r in cells(out)}
    for g, name in enumerate(filters):
        key = "[" + ", ".join(name.split(";")) + "]" if name else "[]"
        assert got[key][1] == cr[g]
        assert got[key][0] == pytest.approx(s[g] / cn[g], rel=1e-6)
    # without the session flag `info` is a string column: the query must be rejected, not silently answered
    r = run(f"SELECT filter, AVG(qual), COUNT(*) FROM vcf_scan('{p}') WHERE info.\"AF\" > 0.01 GROUP BY filter", ok=False)
    assert r.returncode != 0


@pytest.mark.gpu
def test_cli_fastq_histogram(oracle):
    from oracle import decode
    out = run(f"SELECT * FROM fastq_quality_histogram('{FX}/fastq/test.fastq')").stdout
    recs = decode.decode_fastq(f"{FX}/fastq/test.fastq")
    off, data = decode.fastq_device_columns(recs)
    h, _ = oracle.c5_qual_pos_hist(off, data, 64)
    got = {(int(r[0]), int(r[1])): int(r[2]) for r in cells(out)}
    want = {(p + 1, b - 33): int(h[p, b]) for p in range(64) for b in range(256) if h[p, b]}
    assert got == want


@pytest.mark.gpu
def test_cli_big_file_parallel_decode_raw_handoff(tmp_path, oracle):
    """> 8 MB of text: multi-threaded decode, slabs handed to the stream as raw vectors, several staging flushes."""
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    if not os.path.exists(gen):
        subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "gen_text.cpp"), "-o", gen])
    n = 1_500_000
    p = tmp_path / "syn.vcf"
    subprocess.check_call([gen, "vcf", str(n), str(p)])
    assert p.stat().st_size > (8 << 20)
    env = dict(os.environ, EXON_HIP_COALESCE_ROWS="300000")
    r = subprocess.run([CLI, "-q", "-c", "SET exon.vcf_parse_info = true;"
                        f"CREATE EXTERNAL TABLE v STORED AS VCF LOCATION '{p}';"
                        'SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" > 0.01 GROUP BY filter'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    af, av, q, qv, fid = oracle.gen_c4(4, 0, n)
    filters = oracle.c4_filters()
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(af, av, q, qv, fid, filters, 0.01, ">")
    got = {row[0]: (float(row[1]), int(row[2])) for row in cells(r.stdout)}
    for g, name in enumerate(filters):
        key = "[" + ", ".join(name.split(";")) + "]" if name else "[]"
        assert got[key][1] == cr[g]
        assert got[key][0] == pytest.approx(s[g] / cn[g], rel=1e-6)
    out = run(f"SELECT COUNT(*) FROM vcf_scan('{p}') WHERE chrom = '1' AND pos >= 1000 AND pos <= 1200000").stdout
    assert last_count(out) == 1200000 - 1000 + 1


@pytest.mark.gpu
def test_cli_gpu_parse_matches_host_decode(tmp_path):
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    p = tmp_path / "syn.vcf"
    subprocess.check_call([gen, "vcf", "800000", str(p)])
    sql = ["SET exon.vcf_parse_info = true;" f"CREATE EXTERNAL TABLE v STORED AS VCF LOCATION '{p}';"
           'SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" >= 0.25 GROUP BY filter',
           f"SELECT COUNT(*) FROM vcf_scan('{p}') WHERE chrom = '1' AND pos >= 5 AND pos <= 700000"]
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, EXON_HIP_GPU_PARSE=flag)
        r = subprocess.run([CLI, "-q", "-c", *sql], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(sorted(tuple(row) for row in cells(r.stdout) + [["k2", str(last_count(r.stdout))]]))
    assert outs[0] == outs[1]
