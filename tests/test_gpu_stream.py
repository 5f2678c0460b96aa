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
    h = C.c_void_p()  # (ABI 5: the collectives take the library's handle; a caller's own ncclComm_t is wrapped)
    assert ctx.lib.exon_hip_comm_wrap_rccl(comm, C.byref(h)) == 0
    st.all_reduce(h.value)
    st.sync()
    counts, sums = st.finish()
    haf, hav, hq, hqv, hfid = oracle.gen_c4(4, 0, n)
    s_, cn, cr, _ = oracle.c4_cmp_avg_by_group(haf, hav, hq, hqv, hfid, oracle.c4_filters(), 0.01, ">")
    assert np.array_equal(np.array(counts[:5]), cn) and np.array_equal(np.array(counts[5:10]), cr)
    assert np.allclose(np.array(sums), s_, rtol=1e-9)
    st.close()
    plan.close()
    ctx.lib.exon_hip_rccl_comm_destroy(h)
    rccl.ncclCommDestroy(comm)


# ---- ABI 2: overwrite launches, stream reset, gather + fold merge --------------------------------------------------
def _c4_device(ctx, n, seed=4, lo=0):
    af, av, q, qv, fid = ctx.gen_c4(seed, lo, lo + n)
    return [(af, av, None), (q, qv, None), (fid, None, None)]


@pytest.mark.parametrize("n", [0, 1, 2049, 1_000_003, 20_000_000])
def test_plan_launch_overwrite_equals_zero_then_accumulate(ctx, oracle, n):
    """EXON_HIP_LAUNCH_OVERWRITE defines the state whatever it held before; ACCUMULATE adds to it."""
    cols = _c4_device(ctx, max(n, 16))
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5)
    V = plan.n_i64 + plan.n_f64
    garbage = np.full(V, 0x7B7B7B7B7B7B7B7B, np.int64)
    st = ctx.to_device(garbage)
    plan.launch(cols, n, st, overwrite=True)
    ctx.sync()
    got = st.to_host()
    if n:
        haf, hav, hq, hqv, hfid = oracle.gen_c4(4, 0, max(n, 16))
        s, cn, cr, _ = oracle.c4_cmp_avg_by_group(haf[:n], hav, hq[:n], hqv, hfid[:n], oracle.c4_filters(), 0.01, ">")
    else:
        s, cn, cr = np.zeros(5), np.zeros(5, np.int64), np.zeros(5, np.int64)
    assert np.array_equal(got[:5], cn) and np.array_equal(got[5:10], cr)
    assert np.allclose(got[10:].view(np.float64), s, rtol=1e-6, atol=0)
    plan.launch(cols, n, st, overwrite=False)  # accumulate on top: exactly twice the counts
    ctx.sync()
    got2 = st.to_host()
    assert np.array_equal(got2[:10], 2 * got[:10])
    assert np.allclose(got2[10:].view(np.float64), 2 * got[10:].view(np.float64), rtol=1e-12, atol=0)
    plan.close()


def test_plan_launch_overwrite_every_kind(ctx, oracle):
    n = 300_001
    # K2
    c, p = ctx.gen_c2(2, n)
    plan = ctx.plan_region_count(6, 50_000_000, 100_000_000)
    st = ctx.to_device(np.full(1, -5, np.int64))
    plan.launch([(c, None, None), (p, None, None)], n, st, overwrite=True)
    hc, hp = oracle.gen_c2(2, n)
    assert st.to_host()[0] == oracle.c2_region_count(hc, hp, oracle.c2_contigs(), "7:50000000-100000000")[0]
    plan.close()
    # K3
    f, mq, mv, ref, rv = ctx.gen_c3(3, 0, n)
    plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, 25)
    st = ctx.to_device(np.full(26, 77, np.int64))
    plan.launch([(f, None, None), (mq, mv, None), (ref, rv, None)], n, st, overwrite=True)
    hf, hmq, hmv, href, hrv = oracle.gen_c3(3, 0, n)
    want, _ = oracle.c3_flag_mapq_group_count(hf, hmq, hmv, href, hrv, oracle.c3_refs(), 1284, 0, 30)
    assert np.array_equal(st.to_host(), want)
    plan.close()
    # K5: two batches, the first overwrites, the second accumulates
    L, m = 100, 40_000
    off, data = ctx.gen_c5(5, 0, m, L)
    plan = ctx.plan_qual_pos_hist(L)
    st = ctx.to_device(np.full(L * 256, 9, np.int64))
    plan.launch([(data, None, off)], m, st, overwrite=True)
    plan.launch([(data, None, off)], m, st, overwrite=False)
    hoff, hdata = oracle.gen_c5(5, 0, m, L)
    want, _ = oracle.c5_qual_pos_hist(hoff, hdata, L)
    assert np.array_equal(st.to_host().reshape(L, 256), 2 * want)
    plan.close()
    # K6
    ref, rv, s_, e_, pv = ctx.gen_c6(6, 0, n)
    plan = ctx.plan_overlap_count(6, 50_000_000, 100_000_000, columns=(0, 1, 2))
    st = ctx.to_device(np.full(1, 123456, np.int64))
    plan.launch([(ref, rv, None), (s_, pv, None), (e_, pv, None)], n, st, overwrite=True)
    href, hrv, hs, he, hpv = oracle.gen_c6(6, 0, n)
    names = [oracle.c3_refs()[i] for i in range(25)]
    assert st.to_host()[0] == oracle.c6_overlap_count(href, hrv, hs, hpv, he, hpv, names, names[6] + ":50000000-100000000")
    plan.close()


def test_stream_reset_starts_a_new_query_without_a_zeroing_pass(ctx, oracle):
    n = 500_000
    cols = _c4_device(ctx, n)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5)
    st = plan.open()
    st.push_device(cols, n)
    st.push_device(cols, n)
    first, _ = st.finish()
    st.reset()                 # finished stream, new query
    st.push_device(cols, n)
    second, sums = st.finish()
    assert np.array_equal(2 * second, first)
    st.reset()                 # a reset that nothing follows leaves zeros
    third, sums3 = st.finish()
    assert not third.any() and not sums3.any()
    st.close()
    plan.close()


def test_fold_states_is_the_rank_ordered_sum(ctx):
    import ctypes as C
    rng = np.random.default_rng(7)
    world, n_i64, n_f64 = 8, 10, 5
    V = n_i64 + n_f64
    g = np.empty((world, V), np.int64)
    g[:, :n_i64] = rng.integers(0, 1 << 40, (world, n_i64))
    f = rng.standard_normal((world, n_f64)) * 10.0 ** rng.integers(-8, 8, (world, n_f64))
    g[:, n_i64:] = f.view(np.int64)
    d_g, d_o = ctx.to_device(g.reshape(-1)), ctx.empty(np.int64, V)
    ctx._check(ctx.lib.exon_hip_fold_states(ctx.h, None, d_g.ptr, world, n_i64, n_f64, d_o.ptr))
    ctx.sync()
    out = d_o.to_host()
    assert np.array_equal(out[:n_i64], g[:, :n_i64].sum(0))
    acc = np.zeros(n_f64)
    for r in range(world):
        acc = acc + f[r]
    assert np.array_equal(out[n_i64:].view(np.float64), acc)  # same association order: same bits


def test_native_merge_through_the_abi_made_communicator(ctx, oracle):
    """exon_hip_rccl_unique_id / _comm_init / exon_hip_merge_states / _comm_destroy: a one-rank world (one GPU here), the
    call path of the N-GPU merge (ncclAllGather of the packed state on the caller's stream + fold)."""
    import ctypes as C
    uid = (C.c_uint8 * 128)()
    ctx._check(ctx.lib.exon_hip_rccl_unique_id(uid))
    comm = C.c_void_p()
    ctx._check(ctx.lib.exon_hip_rccl_comm_init(ctx.h, uid, 1, 0, C.byref(comm)))
    n = 1_000_000
    cols = _c4_device(ctx, n)
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5)
    V = plan.n_i64 + plan.n_f64
    st, gat, out = ctx.empty(np.int64, V), ctx.empty(np.int64, V), ctx.empty(np.int64, V)
    plan.launch(cols, n, st, overwrite=True)
    ctx._check(ctx.lib.exon_hip_merge_states(ctx.h, None, comm, st.ptr, plan.n_i64, plan.n_f64, gat.ptr, out.ptr))
    ctx.sync()
    assert np.array_equal(out.to_host(), st.to_host())
    ctx._check(ctx.lib.exon_hip_rccl_comm_destroy(comm))
    plan.close()


def test_back_to_back_overwrite_launches_are_bit_identical(ctx):
    """400 launches of the big and the small shape queued without a host sync in between (the finalize of launch i reads
    the workspace that launch i+1 rewrites): all bit-identical -- f64 sums included (fixed fold order)."""
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5)
    V = plan.n_i64 + plan.n_f64
    for n in (6_000_000, 700_001):
        cols = _c4_device(ctx, n)
        st = ctx.empty(np.int64, V)
        plan.launch(cols, n, st, overwrite=True)
        ctx.sync()
        want = st.to_host()
        assert want[5:10].sum() > 0
        outs = [ctx.empty(np.int64, V) for _ in range(400)]
        for o in outs:  # back to back, no host sync in between
            plan.launch(cols, n, o, overwrite=True)
        ctx.sync()
        for o in outs:
            assert np.array_equal(o.to_host(), want)
    plan.close()


def test_small_batches_are_held_and_copied_by_the_pool(ctx, oracle, monkeypatch):
    """Batches of up to 131072 rows are HELD by the stream (released after their slot's pooled copy, not inside the push): many
    8192-row pushes with runs handed to the pool while more arrive, a first stretch without validity bitmaps and a later one with
    (the bitmap starts mid-slot), a big batch in between (copied at once, in order), a reset that drops held batches uncopied, a
    slot boundary inside the held rows -- same state as one big launch over all rows, and as with holding switched off."""
    import pyarrow as pa
    import subprocess, sys
    n = 1_200_000
    af, av, q, qv, fid = oracle.gen_c4(9, 0, n)
    avb, qvb = bits(av, n), bits(qv, n)
    avb[:300_000] = True  # no NULL in the first stretch: those batches carry no bitmap at all
    qvb[:300_000] = True
    av2, qv2 = np.packbits(avb, bitorder="little"), np.packbits(qvb, bitorder="little")
    filters = oracle.c4_filters()

    def rb(lo, hi):
        a = pa.array(af[lo:hi], mask=~avb[lo:hi]) if not avb[lo:hi].all() else pa.array(af[lo:hi])
        b = pa.array(q[lo:hi], mask=~qvb[lo:hi]) if not qvb[lo:hi].all() else pa.array(q[lo:hi])
        return pa.record_batch({"af": a, "qual": b, "filter": pa.DictionaryArray.from_arrays(pa.array(fid[lo:hi]), pa.array(filters))})

    monkeypatch.setenv("EXON_HIP_COALESCE_ROWS", "500000")
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 5)
    st = plan.open()
    for lo in range(0, 90_000, 8192):      # a query that is abandoned: its held batches are released uncopied
        st.push(rb(lo, min(lo + 8192, 90_000)))
    st.reset()
    lo = 0
    while lo < n:
        hi = min(n, lo + (200_000 if lo == 40 * 8192 else 8192))  # one batch above the holding limit in between
        st.push(rb(lo, hi))
        lo = hi
    counts, sums = st.finish()
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(af, av2, q, qv2, fid, filters, 0.01, ">")
    assert np.array_equal(counts[:5], cn) and np.array_equal(counts[5:], cr)
    assert np.allclose(sums, s, rtol=1e-6, atol=0)
    st.close()
    plan.close()
