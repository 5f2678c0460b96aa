"""CRAM 3.0 front end (host decoder -> the BAM device layout -> K3 / K6): SURVEY section 8(f-4).

Pins: the reference's first rows (exon-core/tests/sqllogictests/slt/cram-select-tests.slt:9-12, 33-36, 53-56: r000 / match /
read1-1), the record counts of the container headers, and -- for everything else -- the oracle's independent CRAM decoder
(oracle/decode.py decode_cram: containers, compression header, raw / gzip / rANS 4x8 blocks, data-series encodings, read
features), which the product decoder must equal column for column on every record of the four fixtures."""
import os
import struct

import numpy as np
import pyarrow as pa
import pytest

import exon_amd
from oracle import decode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures", "cram")
FILES = ["test_input_1_a.cram", "0500_mapped.cram", "twolib.sorted.cram", "1404_index_multislice.cram"]


def product_columns(path, region=None):
    scan = exon_amd.Scan(path, "cram", region=region)
    names = scan.dictionary(2)
    batches = [pa.RecordBatch.from_struct_array(b) if isinstance(b, pa.StructArray) else b for b in scan]
    scan.close()
    if not batches:
        return names, [], [], [], [], []
    t = pa.Table.from_batches(batches)
    return (names,) + tuple(t.column(i).to_pylist() for i in range(5))


def test_oracle_reproduces_the_reference_pins():
    refs, recs = decode.decode_cram(os.path.join(FX, "test_input_1_a.cram"))
    r = recs[0]  # cram-select-tests.slt:9-12: r000 99 insert 50 59 30 10M
    assert (r["name"], r["flag"], refs[r["ref_id"]][0], r["start"], r["end"], r["mapq"], r["cigar"]) == ("r000", 99, "insert", 50, 59, 30, "10M")
    assert len(recs) == 15 and recs[-1]["flag"] == 4 and recs[-1]["ref_id"] is None and recs[-1]["start"] is None
    # the SAM-specification example alignments this file is built from
    by_name = {(x["name"], x["flag"]): x for x in recs}
    assert by_name[("r004", 0)]["cigar"] == "6M14N1I5M" and by_name[("r004", 0)]["end"] == 40
    assert by_name[("r003", 0)]["cigar"] == "5H6M" and by_name[("r001", 83)]["cigar"] == "9M"
    refs, recs = decode.decode_cram(os.path.join(FX, "0500_mapped.cram"))
    r = recs[0]  # :33-36: match 99 CHROMOSOME_I 1000 1099
    assert (r["name"], r["flag"], refs[r["ref_id"]][0], r["start"], r["end"]) == ("match", 99, "CHROMOSOME_I", 1000, 1099)
    refs, recs = decode.decode_cram(os.path.join(FX, "twolib.sorted.cram"))
    r = recs[0]  # :53-56: read1-1 0 rand1k 1 60 60 60M
    assert (r["name"], r["flag"], refs[r["ref_id"]][0], r["start"], r["end"], r["mapq"], r["cigar"]) == ("read1-1", 0, "rand1k", 1, 60, 60, "60M")


def container_record_counts(path):
    b = open(path, "rb").read()
    o, counts, first = 26, [], True
    while o < len(b):
        length, = struct.unpack_from("<i", b, o)
        o += 4
        vals = []
        for _ in range(4):
            v, o = decode._itf8(b, o)
            vals.append(v)
        _, o = decode._ltf8(b, o)
        _, o = decode._ltf8(b, o)
        _, o = decode._itf8(b, o)
        nl, o = decode._itf8(b, o)
        for _ in range(nl):
            _, o = decode._itf8(b, o)
        o += 4 + length
        if not first:
            counts.append(vals[3])
        first = False
    return counts


@pytest.mark.parametrize("name", FILES)
def test_host_decoder_equals_the_oracle_on_every_record(name):
    path = os.path.join(FX, name)
    refs, recs = decode.decode_cram(path)
    names, flag, mapq, ref, start, end = product_columns(path)
    assert names == [r[0] for r in refs]
    assert len(flag) == len(recs) == sum(container_record_counts(path))
    assert flag == [r["flag"] for r in recs]
    assert mapq == [r["mapq"] for r in recs]
    assert ref == [None if r["ref_id"] is None else names[r["ref_id"]] for r in recs]
    assert start == [r["start"] for r in recs]
    assert end == [r["end"] for r in recs]


@pytest.mark.parametrize("region", ["CHROMOSOME_I:100-150", "CHROMOSOME_II", "CHROMOSOME_III:1-5", "CHROMOSOME_I:1000000-1000100", "nope:1-2"])
def test_region_filter_is_the_range_hit_of_the_bam_stream(region):
    """cram_region_filter / the indexed CRAM stream keep a record when it overlaps the region on the same reference
    (exon-cram/src/indexed_async_batch_stream.rs, same predicate as exon-bam/src/indexed_async_batch_stream.rs:66-87)."""
    path = os.path.join(FX, "1404_index_multislice.cram")
    refs, recs = decode.decode_cram(path)
    name, _, span = region.partition(":")
    a, b = (1, 2**62) if not span else (int(span.split("-")[0]), int(span.split("-")[1]))
    want = [r for r in recs if r["ref_id"] is not None and refs[r["ref_id"]][0] == name and r["start"] is not None and r["start"] <= b and r["end"] >= a]
    _, flag, _, ref, start, end = product_columns(path, region=region)
    assert len(flag) == len(want) and start == [r["start"] for r in want] and end == [r["end"] for r in want]


def test_malformed_inputs_are_errors(tmp_path):
    good = open(os.path.join(FX, "test_input_1_a.cram"), "rb").read()
    cases = {"magic": b"CRAX" + good[4:], "version": good[:4] + b"\x02\x01" + good[6:], "truncated": good[:1500],
             "short": good[:20], "container_length": good[:26] + struct.pack("<i", 10**9) + good[30:]}
    # an rANS block whose frequency table is cut short / corrupt bytes in the middle of the data containers
    bad = bytearray(good)
    for i in range(1960, 2000):
        bad[i] ^= 0x5A
    cases["corrupt_block"] = bytes(bad)
    for name, data in cases.items():
        p = tmp_path / f"{name}.cram"
        p.write_bytes(data)
        with pytest.raises(exon_amd.ExonHipError):
            scan = exon_amd.Scan(str(p), "cram")
            list(scan)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_cram_file_to_gpu_aggregates_equal_the_oracle(ctx, oracle, name):
    """file -> host CRAM decoder -> staging -> HBM -> K3 (flag / MAPQ predicate, COUNT(*) GROUP BY reference) and K6 (range hit)"""
    from oracle_expect import k3_expected, k6_expected
    path = os.path.join(FX, name)
    for qmin in (0, 30):
        scan = exon_amd.Scan(path, "cram")
        refs = scan.dictionary(2)
        plan = ctx.plan_flag_mapq_group_count(1284, 0, qmin, len(refs), columns=(0, 1, 2))
        st = plan.open()
        rows = st.consume(scan)
        counts, _ = st.finish()
        st.close(); plan.close(); scan.close()
        rows_o, want = k3_expected(oracle, path, "cram", qmin=qmin)
        assert rows == rows_o and np.array_equal(np.array(counts), want)
    for a, b in ((1, None), (5, 60), (1000, 1100)):
        scan = exon_amd.Scan(path, "cram")
        plan = ctx.plan_overlap_count(0, a, b)
        st = plan.open()
        rows = st.consume(scan)
        counts, _ = st.finish()
        st.close(); plan.close(); scan.close()
        assert (rows, int(counts[0])) == k6_expected(path, "cram", refs[0], a, b)


def test_synthetic_cram_with_every_span_changing_feature(tmp_path):
    """Files from tests/cram_writer.py: many containers, two slices each, single- and multi-reference slices, absolute and
    delta positions, detached mates, BETA / multi-symbol HUFFMAN series in the core bit stream, gzip and raw blocks,
    insertions / deletions / soft clips / reference skips / padding / hard clips.  Oracle == truth on 20 k records;
    product == truth on 200 k; pushed-down regions == brute force (and skip whole containers)."""
    from cram_writer import synthetic_records, write_cram
    refs = [("chrA", 3_000_000), ("chrB", 1_500_000), ("chrC", 400_000)]
    for n, per_slice, check_oracle in ((20_000, 300, True), (200_000, 1500, False)):
        recs = synthetic_records(n, refs, seed=n)
        path = str(tmp_path / f"syn{n}.cram")
        write_cram(path, refs, recs, per_slice=per_slice, slices_per_container=2, seed=n)
        want_flag = [r["flag"] for r in recs]
        want_ref = [None if r["ref_id"] < 0 else refs[r["ref_id"]][0] for r in recs]
        want_start = [r["pos"] if r["pos"] > 0 else None for r in recs]
        want_end = [r["pos"] + r["span"] - 1 if r["pos"] > 0 else None for r in recs]
        want_mapq = [None if r["flag"] & 4 or r["mapq"] == 255 else r["mapq"] for r in recs]
        if check_oracle:
            orefs, orecs = decode.decode_cram(path)
            assert orefs == refs and len(orecs) == n
            assert [r["flag"] for r in orecs] == want_flag and [r["start"] for r in orecs] == want_start
            assert [r["end"] for r in orecs] == want_end and [r["mapq"] for r in orecs] == want_mapq
            assert [r["name"] for r in orecs] == [r["name"] for r in recs]
        names, flag, mapq, ref, start, end = product_columns(path)
        assert names == [r[0] for r in refs]
        assert flag == want_flag and ref == want_ref and start == want_start and end == want_end and mapq == want_mapq
        for region, (rname, a, b) in (("chrB:500000-600000", ("chrB", 500_000, 600_000)), ("chrC", ("chrC", 1, 2**62)),
                                      ("chrA:2999000-3000000", ("chrA", 2_999_000, 3_000_000))):
            hit = [i for i in range(n) if want_ref[i] == rname and want_start[i] is not None and want_start[i] <= b and want_end[i] >= a]
            _, f2, _, _, s2, e2 = product_columns(path, region=region)
            assert s2 == [want_start[i] for i in hit] and e2 == [want_end[i] for i in hit] and f2 == [want_flag[i] for i in hit]


def test_bzip2_and_lzma_blocks(tmp_path):
    """Block compression methods 2 (bzip2) and 3 (lzma / xz) -- what htslib writes with use_bzip2 / use_lzma and in its archive
    profile; the reference reads them through noodles-cram's bzip2 / xz dependencies.  The product binds the system's
    libbz2.so.1.0 / liblzma.so.5 at first use; the writer and the oracle use Python's bz2 / lzma modules (independent code).
    Product = oracle = the records written; a damaged payload is an error."""
    from cram_writer import synthetic_records, write_cram
    refs = [("chrA", 3_000_000), ("chrB", 1_500_000), ("chrC", 400_000)]
    n = 20_000
    recs = synthetic_records(n, refs, seed=9)
    want_flag = [r["flag"] for r in recs]
    want_start = [r["pos"] if r["pos"] > 0 else None for r in recs]
    want_end = [r["pos"] + r["span"] - 1 if r["pos"] > 0 else None for r in recs]
    want_mapq = [None if r["flag"] & 4 or r["mapq"] == 255 else r["mapq"] for r in recs]
    raw = None
    for methods in ((2,), (3,), (0, 1, 2, 3)):
        path = str(tmp_path / f"m{'_'.join(map(str, methods))}.cram")
        write_cram(path, refs, recs, per_slice=900, slices_per_container=2, seed=5, methods=methods)
        raw = open(path, "rb").read()
        orefs, orecs = decode.decode_cram(path)
        assert orefs == refs and [r["flag"] for r in orecs] == want_flag and [r["end"] for r in orecs] == want_end
        names, flag, mapq, ref, start, end = product_columns(path)
        assert flag == want_flag and start == want_start and end == want_end and mapq == want_mapq, methods
    # flipped bytes inside bzip2 / lzma payloads: the block CRC (or the codec) must refuse the file, never mis-decode it
    refused = 0
    for at in range(len(raw) // 3, len(raw) - 64, len(raw) // 23):
        bad = bytearray(raw)
        bad[at] ^= 0x55
        p = tmp_path / "bad.cram"
        p.write_bytes(bytes(bad))
        try:
            _, flag, mapq, _, start, end = product_columns(str(p))
            assert flag == want_flag and start == want_start and end == want_end  # the flip fell outside what is read
        except Exception:
            refused += 1
    assert refused >= 10


def test_hostile_compression_headers_and_flipped_bytes(tmp_path):
    """(1) BYTE_ARRAY_LEN encodings nested tens of thousands deep must be an error, not a stack overflow;
    (2) an encoding this reader does not implement (GOLOMB) is an error only when its series is actually read; (3) every
    block carries a CRC-32 that is verified, so a flipped byte in a RAW external block cannot become wrong columns."""
    from cram_writer import IDS, enc_external, enc_len, itf8, synthetic_records, write_cram
    refs = [("chrA", 3_000_000)]
    recs = synthetic_records(3000, refs, seed=3)

    def rows(path):
        scan = exon_amd.Scan(str(path), "cram")
        n = sum(len(b) for b in scan)
        scan.close()
        return n
    good = tmp_path / "good.cram"
    write_cram(str(good), refs, recs, per_slice=500)
    assert rows(good) == 3000
    # (1) nesting: two levels is already outside the specification; 30 000 levels would need ~10 MB of stack
    nested = enc_external(IDS["SC"])
    for _ in range(30_000):
        nested = enc_len(enc_external(IDS["LEN"]), nested)
    for name, enc in (("two", enc_len(enc_len(enc_external(IDS["LEN"]), enc_external(IDS["SC"])), enc_external(IDS["SC"]))), ("deep", nested)):
        p = tmp_path / f"nested_{name}.cram"
        write_cram(str(p), refs, recs[:600], per_slice=500, ds_patch={"SC": enc})
        with pytest.raises(exon_amd.ExonHipError, match="nested"):
            rows(p)
    # (2) GOLOMB (codec 2) on a series nobody reads (TC) is fine; on MQ it is the error
    golomb = itf8(2) + itf8(2) + itf8(0) + itf8(3)
    p = tmp_path / "unused_series.cram"
    write_cram(str(p), refs, recs, per_slice=500, ds_patch={"TC": golomb})
    assert rows(p) == 3000
    p = tmp_path / "used_series.cram"
    write_cram(str(p), refs, recs, per_slice=500, ds_patch={"MQ": golomb})
    with pytest.raises(exon_amd.ExonHipError, match="encoding 2 is not supported"):
        rows(p)
    # (3) flip one payload byte of every kilobyte in turn until a RAW block is hit: always an error, never other columns
    data = bytearray(good.read_bytes())
    flipped = 0
    for at in range(600, len(data) - 64, 997):
        b = bytearray(data)
        b[at] ^= 0x01
        q = tmp_path / "flip.cram"
        q.write_bytes(bytes(b))
        with pytest.raises(exon_amd.ExonHipError):
            rows(q)
        flipped += 1
    assert flipped > 20


def test_corrupted_files_never_crash_the_reader(tmp_path):
    """800 random corruptions of the four fixtures (1-16 bytes overwritten, some files cut short): every one is either decoded
    or reported as an error -- no crash, no hang, no out-of-bounds read (sizes are checked against the bytes in hand)."""
    import random
    rnd = random.Random(7)
    ok = err = 0
    for name in FILES:
        good = open(os.path.join(FX, name), "rb").read()
        for _ in range(200):
            b = bytearray(good)
            for _ in range(rnd.choice([1, 1, 2, 4, 16])):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
            if rnd.random() < 0.1:
                b = b[:rnd.randrange(len(b))]
            p = tmp_path / "fz.cram"
            p.write_bytes(bytes(b))
            try:
                scan = exon_amd.Scan(str(p), "cram")
                sum(len(x) for x in scan)
                scan.close()
                ok += 1
            except exon_amd.ExonHipError:
                err += 1
    assert ok + err == 800 and err > 400


@pytest.mark.parametrize("name", ["1404_index_multislice.cram", "twolib.sorted.cram"])
def test_slice_spans_equal_the_htslib_written_index(name):
    """The reference's fixtures come with the .crai files htslib wrote beside them (exon-core/test-data/datasources/cram/
    1404_index_multislice.cram.crai, two-cram/twolib.sorted.cram.crai; read by exon-core/src/datasources/cram/index.rs:29-41):
    one line per slice and reference -- reference id, alignment start, alignment SPAN, byte offset of the container header,
    offset of the slice header behind it.  Seventeen values of a third party for what this path computes from the read features
    (end = start + reference span - 1): per slice, min(start) and max(end) - min(start) + 1 of the oracle's records AND of the
    product's columns must reproduce every line, as must the container and slice offsets the oracle walked."""
    import gzip
    path = os.path.join(FX, name)
    crai = [tuple(int(x) for x in line.split("\t")) for line in gzip.open(path + ".crai", "rt").read().split("\n") if line]
    refs, recs = decode.decode_cram(path)
    _, flag, mapq, ref, start, end = product_columns(path)
    assert len(flag) == len(recs)
    names = [r[0] for r in refs]
    for source in ("oracle", "product"):
        groups = {}
        for i, r in enumerate(recs):
            if source == "oracle":
                rid, s, e = r["ref_id"], r["start"], r["end"]
            else:
                rid, s, e = (None if ref[i] is None else names.index(ref[i])), start[i], end[i]
            key = (r["container"], r["slice"], -1 if rid is None else rid)
            g = groups.setdefault(key, [None, None])
            if s is not None:
                g[0] = s if g[0] is None else min(g[0], s)
                g[1] = e if g[1] is None else max(g[1], e)
        mine = [(k[2], v[0] or 0, v[1] - v[0] + 1 if v[0] else 1, k[0], k[1]) for k, v in groups.items()]
        assert mine == [c[:5] for c in crai], source


def test_a_child_moved_out_of_a_batch_outlives_its_parent():
    """Arrow C data interface: a consumer may MOVE a child out of a struct array and release the parent first.  CRAM batches
    keep all their buffers in one pooled block; the parent and every child hold a reference to it, so the block returns to
    the pool only when the last of them is released (not when the parent is: a decode thread would reuse it under the child)."""
    import ctypes as C
    from exon_amd import _lib as L
    path = os.path.join(FX, "1404_index_multislice.cram")
    _, recs = decode.decode_cram(path)
    scan = exon_amd.Scan(path, "cram", batch_size=64)  # several small batches: released blocks are taken again at once
    first = scan.next_raw()
    n = first.length
    assert first.n_children == 5 and 0 < n <= 64
    src = first.children[0].contents           # `flag`
    moved = L.ArrowArray()
    C.memmove(C.byref(moved), C.byref(src), C.sizeof(L.ArrowArray))
    src.release = None                         # the move: the source is marked released, the parent will skip it
    release_t = C.CFUNCTYPE(None, C.POINTER(L.ArrowArray))
    C.cast(first.release, release_t)(C.byref(first))
    assert not first.release
    later = []
    while True:                                # the rest of the file: the pool hands out blocks of the same size class
        a = scan.next_raw()
        if a is None:
            break
        later.append(a)
    flags = np.ctypeslib.as_array(C.cast(moved.buffers[1], C.POINTER(C.c_int32)), shape=(n,)).copy()
    assert flags.tolist() == [r["flag"] for r in recs[:n]]
    C.cast(moved.release, release_t)(C.byref(moved))
    assert not moved.release
    for a in later:
        C.cast(a.release, release_t)(C.byref(a))
    scan.close()
