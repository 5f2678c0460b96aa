"""GPU tests of the plan/stream layer: host Arrow batches (Arrow C Data Interface) and HBM-resident batches
(Arrow C Device Data Interface) through exon_hip_stream_*, against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(bm, n):
    return np.unpackbits(bm, bitorder="little")[:n].astype(bool)


def c4_batches(oracle, n, batch=8192, seed=4):
    import pyarrow as pa
    af, av, q, qv, fid = oracle.gen_c4(seed, 0, n)
    avb, qvb = bits(av, n), bits(qv, n)
    filters = oracle.c4_filters()
    tbl = pa.record_batch({
        "chrom": pa.array(np.zeros(n, np.int32)),  # an unused column in front: plans address columns by index
        "af": pa.array(af, mask=~avb), "qual": pa.array(q, mask=~qvb),
        "filter": pa.DictionaryArray.from_arrays(pa.array(fid), pa.array(filters)),
    })
    return (af, av, q, qv, fid), [tbl.slice(i, batch) for i in range(0, n, batch)]


@pytest.mark.parametrize("coalesce", [None, "20000"])
def test_stream_push_host_batches_c4(ctx, oracle, coalesce, monkeypatch):
    if coalesce:
        monkeypatch.setenv("EXON_HIP_COALESCE_ROWS", coalesce)  # force several slot flushes / double buffering
    n = 100_003
    cols, batches = c4_batches(oracle, n)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5, columns=(1, 2, 3))
    st = plan.open()
    for b in batches:  # slices carry non-zero offsets (also inside validity bitmaps)
        st.push(b)
    counts, sums = st.finish()
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(*cols, oracle.c4_filters(), 0.01, ">")
    assert np.array_equal(counts[:5], cn) and np.array_equal(counts[5:], cr)
    assert np.allclose(sums, s, rtol=1e-6, atol=0)
    st.close()
    plan.close()


def test_stream_finish_arrow_state_batch(ctx, oracle):
    n = 30_000
    cols, batches = c4_batches(oracle, n)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5, columns=(1, 2, 3))
    st = plan.open()
    for b in batches:
        st.push(b)
    arr = st.finish_arrow()
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(*cols, oracle.c4_filters(), 0.01, ">")
    rows = arr.to_pylist()
    assert [r["group"] for r in rows] == [g for g in range(5) if cr[g]]
    for r in rows:
        g = r["group"]
        assert r["avg[count]"] == cn[g] and r["count(*)[count]"] == cr[g]
        assert r["avg[sum]"] == pytest.approx(s[g], rel=1e-6)
    assert [f.name for f in arr.type] == ["group", "avg[count]", "avg[sum]", "count(*)[count]"]
    st.close()


def test_stream_push_device_batch(ctx, oracle):
    n = 1_000_000
    af, av, q, qv, fid = ctx.gen_c4(4, 0, n)
    plan = ctx.plan_cmp_avg_by_group(">=", 0.25, 5)
    st = plan.open()
    st.push_device([(af, av, None), (q, qv, None), (fid, None, None)], n)
    st.push_device([(af, av, None), (q, qv, None), (fid, None, None)], n)  # twice: state accumulates
    counts, sums = st.finish()
    haf, hav, hq, hqv, hfid = oracle.gen_c4(4, 0, n)
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(haf, hav, hq, hqv, hfid, oracle.c4_filters(), 0.25, ">=")
    assert np.array_equal(counts[:5], 2 * cn) and np.array_equal(counts[5:], 2 * cr)
    assert np.allclose(sums, 2 * s, rtol=1e-6, atol=0)
    st.close()


def test_stream_region_count_and_bam(ctx, oracle):
    import pyarrow as pa
    n = 250_000
    c, p = oracle.gen_c2(2, n)
    contigs = oracle.c2_contigs()
    rb = pa.record_batch({"chrom": pa.DictionaryArray.from_arrays(pa.array(c), pa.array(contigs)), "pos": pa.array(p)})
    plan = ctx.plan_region_count(contigs.index("7"), 50_000_000, 100_000_000)
    st = plan.open()
    for i in range(0, n, 8192):
        st.push(rb.slice(i, 8192))
    counts, _ = st.finish()
    assert counts[0] == oracle.c2_region_count(c, p, contigs, "7:50000000-100000000")[0]
    st.close()

    f, mq, mv, ref, rv = oracle.gen_c3(3, 0, n)
    refs = oracle.c3_refs()
    rb = pa.record_batch({
        "flag": pa.array(f), "mapq": pa.array(mq, mask=~bits(mv, n)),
        "reference": pa.DictionaryArray.from_arrays(pa.array(np.where(bits(rv, n), ref, 0).astype(np.int32), mask=~bits(rv, n)), pa.array(refs)),
    })
    plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, len(refs))
    st = plan.open()
    for i in range(0, n, 10_000):
        st.push(rb.slice(i, 10_000))
    counts, _ = st.finish()
    want, _ = oracle.c3_flag_mapq_group_count(f, mq, mv, ref, rv, refs, 1284, 0, 30)
    assert np.array_equal(counts, want)
    st.close()


def test_stream_fastq_histogram_from_scan(ctx, oracle, tmp_path):
    """text FASTQ -> native decoder -> stream -> K5, in native code end to end (exon_hip_stream_consume_scan)."""
    import exon_amd
    n, L = 20_000, 100
    off, data = oracle.gen_c5(5, 0, n, L)
    p = tmp_path / "syn.fastq"
    with open(p, "wb") as fh:
        for r in range(n):
            fh.write(b"@r%d\n" % r + b"A" * L + b"\n+\n" + data[r * L:(r + 1) * L].tobytes() + b"\n")
    plan = ctx.plan_qual_pos_hist(L, columns=(3,))
    st = plan.open()
    scan = exon_amd.Scan(p, "fastq")
    assert st.consume(scan) == n
    counts, _ = st.finish()
    want, _ = oracle.c5_qual_pos_hist(off, data, L)
    assert np.array_equal(counts.reshape(L, 256), want)
    st.close()
    scan.close()


def test_push_after_finish_is_an_error(ctx, oracle):
    import exon_amd
    _, batches = c4_batches(oracle, 100)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5, columns=(1, 2, 3))
    st = plan.open()
    st.finish()
    with pytest.raises(exon_amd.ExonHipError):
        st.push(batches[0])
    st.close()


def test_native_rccl_all_reduce_on_a_one_rank_communicator(ctx, oracle):
    """exon_hip_stream_all_reduce: the AggregateExec(Final) merge through librccl itself (no torch).  One GPU, so the
    communicator has one rank and the state must come back unchanged; the call path (dlopen, data types, stream) is what
    the N-GPU merge uses."""
    import ctypes as C
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so")
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    n = 1_000_000
    af, av, q, qv, fid = ctx.gen_c4(4, 0, n)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5, columns=(0, 1, 2))
    st = plan.open()
    st.push_device([(af, av, None), (q, qv, None), (fid, None, None)], n)
    st.all_reduce(comm.value)
    st.sync()
    counts, sums = st.finish()
    haf, hav, hq, hqv, hfid = oracle.gen_c4(4, 0, n)
    s_, cn, cr, _ = oracle.c4_cmp_avg_by_group(haf, hav, hq, hqv, hfid, oracle.c4_filters(), 0.01, ">")
    assert np.array_equal(np.array(counts[:5]), cn) and np.array_equal(np.array(counts[5:10]), cr)
    assert np.allclose(np.array(sums), s_, rtol=1e-9)
    st.close()
    plan.close()
    rccl.ncclCommDestroy(comm)
