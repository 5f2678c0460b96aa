"""CRAM 3.1's rANS Nx16 block codec (method 5) -- htslib's default since 1.22; the reference reads it through noodles-cram
(exon-cram/src/async_batch_stream.rs; the codec itself is not in /root/reference and is restated from the published "CRAM
codecs" specification).  PARITY NOTE: no htslib-written 3.1 stream exists in this image or among the reference's fixtures (all
four are CRAM 3.0), so these tests pin the product decoder (exon_amd/csrc/host/cram.h) on streams of a test-side encoder
(tests/rans_nx16_writer.py) and on the oracle's own Python decoder (oracle/decode.py); what stands in for the missing golden
vector is the rANS end-state test both decoders apply (every state must return to the encoder's initial value), which these
tests show refuses damaged entropy-coded payloads."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rans_nx16_writer as nx  # noqa: E402
from oracle import decode  # noqa: E402
from test_cram import product_columns  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    """the product's decoder compiled from the header the library is built from (no GPU needed)"""
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("nx16") / "rans_nx16_harness")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "exon_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "rans_nx16_harness.cpp"), "-o", exe, "-lz", "-ldl", "-lpthread"], check=True)

    def run(cases):
        """cases: [(expected size, stream)] -> [(ok, bytes)]"""
        inp = b"".join(struct.pack("<II", n, len(s)) + s for n, s in cases)
        r = subprocess.run([exe], input=inp, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        out, o, res = r.stdout, 0, []
        for _ in cases:
            st, n = struct.unpack_from("<iI", out, o)
            res.append((st == 0, out[o + 8:o + 8 + n]))
            o += 8 + n
        assert o == len(out)
        return res
    return run


def _datasets(rng):
    yield b""
    yield b"A"
    yield b"AB" * 3
    yield bytes(rng.integers(0, 256, 1000, dtype=np.uint8))                     # incompressible, all 256 symbols
    yield bytes(rng.integers(0, 3, 5000, dtype=np.uint8))                       # packs into 2 bits
    yield bytes(np.repeat(rng.integers(0, 8, 300, dtype=np.uint8), rng.integers(1, 30, 300)))  # long runs
    yield bytes(rng.choice([0, 1, 2, 60, 255], 4000, p=[.5, .2, .2, .05, .05]).astype(np.uint8))  # mapq-like
    yield b"\x07" * 777                                                         # one symbol
    yield bytes(range(256)) * 3                                                 # every context used once per pass
    yield bytes(rng.integers(0, 256, 37, dtype=np.uint8))                       # shorter than 32 states x 2
    yield bytes((rng.integers(0, 40, 3000) + np.arange(3000) // 100).astype(np.uint8))  # drifting alphabet (order 1 helps)
    yield b"".join(struct.pack(">I", int(v)) for v in rng.integers(0, 1 << 20, 800))    # 4-byte integers: striping helps


ALL_FLAGS = [0x00, 0x01, 0x04, 0x05, 0x20, 0x40, 0x41, 0x44, 0x45, 0x60, 0x80, 0x81, 0xA0, 0xC0, 0xC1, 0xC4, 0xC5, 0x08, 0x09, 0x0C, 0x10, 0x11]


def test_every_flag_combination_round_trips(harness):
    """order 0 / 1, 4 / 32 states, stored, run-length coded (side stream stored or coded), bit-packed, striped, size omitted:
    product decoder == oracle decoder == the bytes that were encoded"""
    rng = np.random.default_rng(7)
    cases = []
    for data in _datasets(rng):
        for flags in ALL_FLAGS:
            for opt in (dict(), dict(o1_bits=10, code_table=True, code_rle_meta=False)):
                cases.append((data, nx.encode(data, flags, **opt)))
    res = harness([(len(d), s) for d, s in cases])
    for (data, stream), (ok, got) in zip(cases, res):
        assert ok and got == data, (len(data), hex(stream[0]), got[:80])
        assert decode._rans_nx16(stream, len(data)) == data
    # entropy coding does compress: the skewed 5-symbol column at order 0 is well under its 4000 bytes
    skew = bytes(np.random.default_rng(1).choice([0, 1, 2, 60, 255], 4000, p=[.5, .2, .2, .05, .05]).astype(np.uint8))
    assert len(nx.encode(skew, 0)) < 2200 and len(nx.encode(skew, nx.PACK)) < 2200


def test_damaged_payloads_are_refused_not_misdecoded(harness):
    """A CRAM block's CRC-32 covers the COMPRESSED bytes; in a stream decoded under a wrong reading of the format (a wrong
    interleave, table layout or renormalisation rule -- what an unpinned restatement risks) nothing else would object.  The
    end-state test does: every kind of damage that makes the decoder lose step with the encoder is refused.  Flip or drop bytes
    inside the states + renormalisation words of order-0 / order-1 streams (4 and 32 states): the decoder reports an error or
    still returns the original bytes.  (What the test cannot see is a flip that turns one symbol into another of the SAME
    frequency at the same offset -- the state does not change, as in a fixed-width code; that is what the block CRC is for.
    Such flips are counted and must stay rare on skewed data.)"""
    rng = np.random.default_rng(11)
    cases, truth = [], []
    skewed = [d for d in _datasets(rng) if len(d) > 100 and len(set(np.bincount(np.frombuffer(d, np.uint8), minlength=256).tolist())) > 4]
    assert len(skewed) >= 5
    for data in skewed:
        for flags in (0x00, 0x01, 0x04, 0x05):
            s = nx.encode(data, flags)
            n_states = 32 if flags & 4 else 4
            lo = len(s) - max(8, min(len(s) // 3, 600))  # inside the words that follow the tables
            lo = max(lo, len(s) - (len(s) - 4 * n_states) // 2)
            for at in rng.integers(lo, len(s), 12):
                bad = bytearray(s)
                bad[int(at)] ^= int(rng.integers(1, 256))
                cases.append((len(data), bytes(bad)))
                truth.append(data)
            cases.append((len(data), s[:-2]))  # a dropped word
            truth.append(data)
            cases.append((len(data), s[:lo]))
            truth.append(data)
    res = harness(cases)
    refused = silent = 0
    for (n, stream), data, (ok, got) in zip(cases, truth, res):
        if not ok:
            refused += 1
        elif got != data:
            silent += 1
            assert sum(a != b for a, b in zip(got, data)) <= 2, "lost step with the encoder and was not refused"
        try:
            py = decode._rans_nx16(stream, n)
            assert py == data or sum(a != b for a, b in zip(py, data)) <= 2
        except (ValueError, IndexError, KeyError, struct.error, TypeError):
            pass
    assert refused >= 0.95 * len(cases) and silent <= 0.02 * len(cases), (refused, silent, len(cases))


def test_hostile_streams_are_errors(harness):
    """sizes and counts that lie: every one is a reported error (the harness process must survive them all)"""
    u7 = nx.uint7
    big = u7(1 << 30)
    cases = [
        (10, b""),                                                   # no flags byte
        (10, bytes([0x00])),                                         # no size
        (10, bytes([0x00]) + big),                                   # absurd size
        (10, bytes([0x20]) + u7(10) + b"abc"),                       # stored, but short
        (10, bytes([0x08]) + u7(10) + bytes([0])),                   # zero stripes
        (10, bytes([0x08]) + u7(10) + bytes([2]) + u7(500) + u7(1)),  # stripe longer than the stream
        (10, bytes([0x08]) + u7(10) + bytes([1]) + u7(4) + bytes([0x18, 1, 0, 0])),  # stripes inside stripes
        (10, bytes([0x80]) + u7(10) + bytes([17]) + bytes(17) + u7(5) + bytes(40)),   # 17 packed symbols
        (10, bytes([0xA0]) + u7(10) + bytes([2, 65, 66]) + u7(1) + bytes([0])),       # 1 packed byte cannot hold 10 symbols
        (10, bytes([0x60]) + u7(10) + u7(2 * 2 + 1) + u7(1) + bytes([1, 65]) + b"A"),  # run lengths missing
        (10, bytes([0x60]) + u7(10) + u7(2 * 3 + 1) + u7(1) + bytes([1, 65, 100]) + b"A"),  # run past the block
        (10, bytes([0x00]) + u7(10) + bytes([65, 0]) + u7(4097) + bytes(16)),         # frequency above 4096
        (10, bytes([0x00]) + u7(10) + bytes([65, 0]) + u7(3000) + bytes(16)),         # frequencies not a power of two
        (10, bytes([0x00]) + u7(10) + bytes([65, 66, 200])),                          # symbol run past 255
        (10, bytes([0x01]) + u7(10) + bytes([0xD0])),                                 # 13-bit order-1 table
        (10, bytes([0x01]) + u7(10) + bytes([0xC1]) + big + u7(3) + bytes(3)),        # order-1 table of 1 GiB
    ]
    good = nx.encode(b"hello hello hello", 0)
    cases.append((16, good))            # decodes, but not to the block's size
    res = harness(cases)
    for (n, s), (ok, msg) in zip(cases, res):
        assert not ok, (s[:12], msg)
        assert msg.startswith(b"CRAM: "), msg


def _truth(recs):
    return ([r["flag"] for r in recs], [r["pos"] if r["pos"] > 0 else None for r in recs],
            [r["pos"] + r["span"] - 1 if r["pos"] > 0 else None for r in recs],
            [None if r["flag"] & 4 or r["mapq"] == 255 else r["mapq"] for r in recs])


def test_cram_31_files(tmp_path):
    """CRAM 3.1 files whose blocks are rANS Nx16 streams with every transform (and mixed with the 3.0 methods): product = oracle =
    the records written, region scans included"""
    from cram_writer import synthetic_records, write_cram
    refs = [("chrA", 3_000_000), ("chrB", 1_500_000), ("chrC", 400_000)]
    recs = synthetic_records(12_000, refs, seed=21)
    want_flag, want_start, want_end, want_mapq = _truth(recs)
    for methods in ((5,), (0, 1, 4 if False else 5, 2)):
        path = str(tmp_path / f"v31_{len(methods)}.cram")
        write_cram(path, refs, recs, per_slice=800, slices_per_container=2, seed=3, methods=methods, qualities=True)
        assert open(path, "rb").read(6) == b"CRAM\x03\x01"
        orefs, orecs = decode.decode_cram(path)
        assert orefs == refs and [r["flag"] for r in orecs] == want_flag and [r["end"] for r in orecs] == want_end
        assert [r["name"] for r in orecs] == [r["name"] for r in recs]
        _, flag, mapq, ref, start, end = product_columns(path)
        assert flag == want_flag and start == want_start and end == want_end and mapq == want_mapq, methods
    a, b = 200_000, 900_000
    hit = [i for i, r in enumerate(recs) if r["ref_id"] == 0 and r["pos"] > 0 and r["pos"] <= b and r["pos"] + r["span"] - 1 >= a]
    _, f2, _, _, s2, e2 = product_columns(path, region=f"chrA:{a}-{b}")
    assert len(hit) > 100 and s2 == [want_start[i] for i in hit] and e2 == [want_end[i] for i in hit] and f2 == [want_flag[i] for i in hit]


def test_blocks_of_discarded_series_in_unsupported_codecs_are_never_opened(tmp_path):
    """htslib's 3.1 profiles code read names with the name tokeniser (method 8) and, in the small / archive profiles, quality
    scores with fqzcomp (7).  This path discards both series, so their blocks stay closed and the file decodes; when a series
    the columns NEED sits in such a block the error names the codec (it is not a mis-decode and not a crash)."""
    from cram_writer import synthetic_records, write_cram
    import exon_amd
    refs = [("chrA", 3_000_000), ("chrB", 1_500_000)]
    recs = synthetic_records(5_000, refs, seed=4)
    want_flag, want_start, want_end, want_mapq = _truth(recs)
    path = str(tmp_path / "tok3_fqz.cram")
    write_cram(path, refs, recs, per_slice=600, seed=2, methods=(5,), qualities=True, opaque={"RN": 8, "QS": 7})
    _, flag, mapq, ref, start, end = product_columns(path)
    assert flag == want_flag and start == want_start and end == want_end and mapq == want_mapq
    orefs, orecs = decode.decode_cram(path)  # the oracle leaves such blocks closed too: names read as None
    assert [r["flag"] for r in orecs] == want_flag and [r["end"] for r in orecs] == want_end and orecs[0]["name"] is None
    for series, method, word in (("BF", 6, "arithmetic"), ("AP", 7, "fqzcomp"), ("FN", 8, "tokeniser")):
        p = str(tmp_path / f"needs_{series}.cram")
        write_cram(p, refs, recs[:900], per_slice=600, seed=2, methods=(5,), opaque={series: method})
        with pytest.raises(exon_amd.ExonHipError, match=word):
            product_columns(p)
    # a version this reader does not know is refused up front
    raw = bytearray(open(path, "rb").read())
    raw[5] = 2
    bad = tmp_path / "v32.cram"
    bad.write_bytes(bytes(raw))
    with pytest.raises(exon_amd.ExonHipError, match="3.2"):
        product_columns(str(bad))


@pytest.mark.gpu
def test_cram_31_file_to_gpu_aggregates(ctx, oracle, tmp_path):
    """a CRAM 3.1 file (rANS Nx16 blocks, names / qualities in codecs nobody here can open) -> host decoder -> staging -> HBM -> K3 /
    K6: the aggregates equal the oracle's over its own decode of the same file"""
    import exon_amd
    from cram_writer import synthetic_records, write_cram
    from oracle_expect import k3_expected, k6_expected
    refs = [("chrA", 3_000_000), ("chrB", 1_500_000), ("chrC", 400_000)]
    recs = synthetic_records(8_000, refs, seed=33)
    path = str(tmp_path / "v31_gpu.cram")
    write_cram(path, refs, recs, per_slice=700, seed=6, methods=(5,), qualities=True, opaque={"RN": 8, "QS": 7})
    for qmin in (0, 30):
        scan = exon_amd.Scan(path, "cram")
        names = scan.dictionary(2)
        plan = ctx.plan_flag_mapq_group_count(1284, 0, qmin, len(names), columns=(0, 1, 2))
        st = plan.open()
        rows = st.consume(scan)
        counts, _ = st.finish()
        st.close(); plan.close(); scan.close()
        rows_o, want = k3_expected(oracle, path, "cram", qmin=qmin)
        assert rows == rows_o == len(recs) and np.array_equal(np.array(counts), want)
    scan = exon_amd.Scan(path, "cram")
    plan = ctx.plan_overlap_count(0, 200_000, 900_000)
    st = plan.open()
    rows = st.consume(scan)
    counts, _ = st.finish()
    st.close(); plan.close(); scan.close()
    assert (rows, int(counts[0])) == k6_expected(path, "cram", "chrA", 200_000, 900_000)
